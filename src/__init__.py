"""Package with the reference's module layout (src.epipolar, src.triangulation, src.util, src.logger, src.plot,
src.display) so that a driver written against the reference imports this implementation unchanged."""
