"""Import-path mirror of the reference's ``Full_model`` package (hot-path files only)."""
