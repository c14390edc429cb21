"""Mirror of hyvideo/modules/__init__.py:4-26: load_model(args, in_channels, out_channels, factor_kwargs)."""
from .models import HYVideoDiffusionTransformer, HUNYUAN_VIDEO_CONFIG


def load_model(args, in_channels, out_channels, factor_kwargs):
    if args.model in HUNYUAN_VIDEO_CONFIG.keys():
        return HYVideoDiffusionTransformer(args, in_channels=in_channels, out_channels=out_channels,
                                           **HUNYUAN_VIDEO_CONFIG[args.model], **factor_kwargs)
    raise NotImplementedError()
