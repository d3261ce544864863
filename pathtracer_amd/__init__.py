"""MI355X-native path-tracing core: Python host mirror of the reference's
SceneLoader / Renderer over the C-ABI in include/ptcore.h."""
from . import _ffi
from .scene import Scene, SceneLoader, override_scene_text
from .renderer import Renderer, resultToPNG

__all__ = ["Scene", "SceneLoader", "Renderer", "resultToPNG", "override_scene_text", "_ffi"]
