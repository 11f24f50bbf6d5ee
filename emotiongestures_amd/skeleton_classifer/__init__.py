"""Import-path mirror of the reference's ``skeleton_classifer`` package."""
