from .track_instances import TrackInstances  # noqa: F401
