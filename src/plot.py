"""Import-safe placeholder for the reference's 2D plotting module (visualisation is outside this package's scope)."""


def plot_image_matches(*args, **kwargs):
    raise NotImplementedError("plotting is not part of the MI355X hot-path package")
