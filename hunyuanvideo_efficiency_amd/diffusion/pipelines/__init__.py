from .pipeline_hunyuan_video import HunyuanVideoPipeline  # noqa: F401
