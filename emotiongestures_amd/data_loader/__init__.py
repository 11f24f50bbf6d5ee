"""Mirrors of the reference's data_loader package for the sample path (see ..datapath)."""
