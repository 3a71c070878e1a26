"""MI355X-native drop-in for the `Video Stabilizer Flow` / `Video Stabilizer Motion Apply` nodes of
nomadoor/ComfyUI-Video-Stabilizer (dense-flow hot path only).

ComfyUI discovers the package through `comfy_entrypoint()` exactly as it does the reference
(__init__.py:37-39 there)."""

__version__ = "0.1.0"


async def comfy_entrypoint():
    from .nodes import VideoStabilizerAmdExtension

    return VideoStabilizerAmdExtension()
