from .pipeline import DeviceSamplePipeline, IterativeRefiner  # noqa: F401
