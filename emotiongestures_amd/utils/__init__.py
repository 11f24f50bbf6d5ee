"""Mirrors of the reference's utils package for the sample path (see ..datapath)."""
