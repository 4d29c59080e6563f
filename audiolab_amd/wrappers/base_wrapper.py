"""Plugin API of the Process tab (reference wrappers/base_wrapper.py): ``TypedInput`` (:26-98) and the
singleton ``BaseWrapper`` (:101-135) with its abstract ``process_audio`` -- the drop-in boundary
(SURVEY 8(b) b1).  Headless: Gradio / FastAPI / pydantic are NOT imported (the reference's UI
rendering, :606-743, and REST plumbing, :248-509, are out of scope); ``TypedInput.field`` keeps the
attribute names the reference reads from ``pydantic.Field`` (default, ge, le, description ...).
"""
from __future__ import annotations

import os
import re
from abc import abstractmethod
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Tuple, Union

from audiolab_amd.util.data_classes import ProjectFiles


class TypedInput:
    def __init__(self, default: Any = ..., description: str = None, ge: float = None, le: float = None, step: float = None,
                 min_length: int = None, max_length: int = None, regex: str = None, choices: List[Union[str, int]] = None,
                 type: type = None, gradio_type: str = None, render: bool = True, required: bool = False,
                 refresh: Callable = None, on_change: Callable = None, on_click: Callable = None,
                 on_select: Callable = None, controls: List[str] = None, group_name: str = None):
        self.field = SimpleNamespace(default=default, description=description, ge=ge, le=le, step=step,
                                     min_length=min_length, max_length=max_length, pattern=regex,
                                     json_schema_extra={"enum": choices} if choices else None, required=required)
        self.type = type
        self.render = render
        self.required = required
        self.refresh = refresh
        self.description = description
        self.choices = choices
        self.on_change, self.on_click, self.on_select = on_change, on_click, on_select
        self.controls = controls
        self.gradio_type = gradio_type if gradio_type else self.pick_gradio_type()
        self.group_name = group_name

    def pick_gradio_type(self):
        """:80-98."""
        if self.type == bool:
            return "Checkbox"
        if self.type == str:
            return "Text"
        if self.type in (int, float) and self.field.ge is not None and self.field.le is not None:
            return "Slider"
        if self.type == float:
            return "Number"
        if self.type == list:
            return "Textfield"
        if self.choices:
            return "Dropdown"
        return "Text"


class BaseWrapper:
    _instance = None
    priority = 1000
    allowed_kwargs: Dict[str, TypedInput] = {}
    description = "Base Wrapper"
    default = False
    required = False
    hidden_groups: List[str] = []

    def __new__(cls):
        """One instance per wrapper class (:110-118); ``title`` is derived from the class name."""
        if cls.__dict__.get("_instance") is None:
            inst = super(BaseWrapper, cls).__new__(cls)
            inst.arg_handler = None
            inst.title = " ".join(w.capitalize() for w in re.sub(r"(?<!^)(?=[A-Z])", "_", cls.__name__).split("_"))
            cls._instance = inst
        return cls._instance

    def validate_args(self, **kwargs: Dict[str, Any]) -> bool:
        filtered = {k: v for k, v in kwargs.items() if k in self.allowed_kwargs}
        for arg, value in self.allowed_kwargs.items():
            if value.required and not filtered.get(arg):
                return False
        return True

    @abstractmethod
    def process_audio(self, inputs: List[ProjectFiles], callback=None, **kwargs: Dict[str, Any]) -> List[ProjectFiles]:
        pass

    @staticmethod
    def filter_inputs(project: ProjectFiles, input_type: str = "audio") -> Tuple[List[str], List[str]]:
        """:745-821 (audio / any file types; video demux is out of scope)."""
        inputs = project.last_outputs
        if not inputs:
            stem_dir = os.path.join(project.project_dir, "stems")
            if os.path.exists(stem_dir):
                voc = [f for f in os.listdir(stem_dir) if "(Vocals)" in f]
                if voc:
                    inputs = [os.path.join(stem_dir, voc[0])]
                else:
                    inputs = [os.path.join(stem_dir, f) for f in os.listdir(stem_dir)
                              if os.path.isfile(os.path.join(stem_dir, f)) and not f.endswith(".json")]
            if not inputs:
                inputs = [project.src_file]
        exts = {"audio": ["mp3", "wav", "flac", "m4a", "aac", "ogg", "opus"], "text": ["txt", "csv", "json"],
                "image": ["jpg", "jpeg", "png", "gif", "bmp", "tiff", "webp"],
                "video": ["mp4", "mov", "avi", "webm", "mkv", "flv"], "any": []}.get(input_type, [])
        keep, rest = [], []
        for f in inputs:
            ext = os.path.splitext(f)[1][1:].lower()
            (keep if (not exts or ext in exts) else rest).append(f)
        return keep, rest
