"""Import-path mirror of the reference's ``CAVE`` package."""
