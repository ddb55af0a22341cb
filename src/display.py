"""Import-safe placeholder for the reference's Pangolin 3D viewer module (outside this package's scope)."""


def setup(*args, **kwargs):
    raise NotImplementedError("3D display is not part of the MI355X hot-path package")
