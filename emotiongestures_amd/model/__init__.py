"""Import-path mirror of the reference's ``model`` package (metric-side files of the hot path's caller only)."""
