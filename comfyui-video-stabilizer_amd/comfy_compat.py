"""ComfyUI runtime surface used by the nodes.

Inside ComfyUI the real `comfy_api.latest`, `comfy.utils.ProgressBar` and
`comfy.model_management` are used.  Outside (tests, bench, smoke) a minimal stand-in with the
same attribute names is provided so the node classes can be imported, their schema inspected
and `execute` called directly -- the same trick the reference's own check scripts use
(scripts/compare_refactor_behavior.py:75-109).
"""

from __future__ import annotations

from typing import Any, List

try:  # pragma: no cover - exercised only inside ComfyUI
    from comfy_api.latest import ComfyExtension, io  # type: ignore
    from comfy.utils import ProgressBar  # type: ignore

    try:
        import comfy.model_management as model_management  # type: ignore
    except ImportError:
        model_management = None
    HAVE_COMFY = True
except ImportError:
    HAVE_COMFY = False
    model_management = None

    class ProgressBar:  # noqa: D401 - same call surface as comfy.utils.ProgressBar
        def __init__(self, total: int):
            self.total = int(total)
            self.current = 0
            self.updates: List[tuple] = []

        def update_absolute(self, value: int, total: int | None = None) -> None:
            self.current = int(value)
            if total is not None:
                self.total = int(total)
            self.updates.append((self.current, self.total))

        def update(self, value: int) -> None:
            self.update_absolute(self.current + int(value))

    class _Socket:
        def __init__(self, kind: str, direction: str, name: str, **options: Any):
            self.kind, self.direction, self.id, self.options = kind, direction, name, options
            self.display_name = options.get("display_name")

        def __repr__(self) -> str:
            return f"{self.kind}.{self.direction}({self.id!r})"

    def _socket_type(kind: str):
        class _T:
            @staticmethod
            def Input(name: str, **options: Any):
                return _Socket(kind, "Input", name, **options)

            @staticmethod
            def Output(name: str | None = None, **options: Any):
                return _Socket(kind, "Output", name, **options)

        _T.__name__ = kind
        return _T

    class _Schema:
        def __init__(self, node_id: str, display_name: str = "", category: str = "", description: str = "", **extra: Any):
            self.node_id, self.display_name, self.category, self.description = node_id, display_name, category, description
            self.inputs: list = []
            self.outputs: list = []
            self.is_deprecated = bool(extra.get("is_deprecated", False))
            self.extra = extra

    class _NodeOutput:
        def __init__(self, *values: Any, **kwargs: Any):
            self.result = tuple(values)
            self.args = tuple(values)
            self.kwargs = kwargs

        def __iter__(self):
            return iter(self.result)

        def __getitem__(self, i):
            return self.result[i]

    class _NumberDisplay:
        slider = "slider"
        number = "number"

    class _ControlAfterGenerate:
        fixed = "fixed"
        increment = "increment"
        decrement = "decrement"
        randomize = "randomize"

    class _IO:
        Schema = _Schema
        NodeOutput = _NodeOutput
        NumberDisplay = _NumberDisplay
        ControlAfterGenerate = _ControlAfterGenerate
        Image = _socket_type("Image")
        Mask = _socket_type("Mask")
        Float = _socket_type("Float")
        Int = _socket_type("Int")
        Boolean = _socket_type("Boolean")
        Combo = _socket_type("Combo")
        Color = _socket_type("Color")
        String = _socket_type("String")

        class ComfyNode:
            @classmethod
            def define_schema(cls):  # pragma: no cover - overridden
                raise NotImplementedError

        @staticmethod
        def Custom(kind: str):
            return _socket_type(kind)

    io = _IO()

    class ComfyExtension:
        async def on_load(self) -> None:
            return None

        async def get_node_list(self) -> list:
            return []


def check_interrupt() -> None:
    """Cooperative cancel polled between kernel batches (flow.py:275-277)."""
    if model_management is not None:
        model_management.throw_exception_if_processing_interrupted()
