"""MI355X-native drop-in for the `Video Stabilizer Flow` / `Video Stabilizer Motion Apply`
nodes of nomadoor/ComfyUI-Video-Stabilizer (hot path only)."""

__version__ = "0.1.0"
