"""``ProjectFiles`` -- the unit of work handed between wrappers.

Contract (what callers of the reference's class rely on, util/data_classes.py:10-67):
  * project directory ``<output_path>/process/<input stem>_<first 8 hex digits of xxh64(file bytes)>/`` with the input
    copied to ``source/<file name>`` once;
  * attributes ``src_file, file_hash, project_dir, last_outputs, video_sources, file_dict, output_dict``;
    ``file_dict`` maps every existing sub-folder name to the files under it (``"source"`` first);
  * ``add_output(process, paths)`` records the paths under both dicts and makes them ``last_outputs``;
  * ``all_outputs()`` -- existing files of every process except merge / convert / export, in order, without repeats.
The attribute set is the interface; the code below is this build's own."""
from __future__ import annotations

import shutil
from pathlib import Path
from typing import Dict, Iterable, List, Union

import xxhash

from audiolab_amd.handlers import config

_TERMINAL_PROCESSES = frozenset({"merge", "convert", "export"})     # their files are not inputs of a later stage


def content_tag(path: Union[str, Path], digits: int = 8, block: int = 1 << 20) -> str:
    """leading hex digits of the xxh64 of the file's bytes"""
    digest = xxhash.xxh64()
    with open(path, "rb") as stream:
        for piece in iter(lambda: stream.read(block), b""):
            digest.update(piece)
    return digest.hexdigest()[:digits]


class ProjectFiles:
    def __init__(self, input_file):
        origin = Path(input_file)
        self.file_hash = content_tag(origin)
        home = Path(config.output_path) / "process" / f"{origin.stem}_{self.file_hash}"
        kept = home / "source" / origin.name
        kept.parent.mkdir(parents=True, exist_ok=True)
        if not kept.exists():
            shutil.copyfile(origin, kept)
        self.src_file = str(kept)
        self.project_dir = str(home)
        self.last_outputs: List[str] = []
        self.video_sources: Dict = {}
        self.output_dict: Dict[str, List[str]] = {}
        self.file_dict: Dict[str, List[str]] = {"source": [self.src_file]}
        self._index_existing(home)

    def _index_existing(self, home: Path) -> None:
        """files left by earlier runs, keyed by the name of the folder that holds them"""
        for folder in sorted(p for p in home.rglob("*") if p.is_dir()):
            bucket = self.file_dict.setdefault(folder.name, [])
            bucket.extend(str(f) for f in sorted(folder.iterdir()) if f.is_file())

    def add_output(self, process: str, outputs: Union[List[str], str]):
        paths = [outputs] if isinstance(outputs, str) else outputs
        self.last_outputs = paths
        for table in (self.file_dict, self.output_dict):
            table.setdefault(process, []).extend(paths)

    def all_outputs(self) -> List[str]:
        def live(paths: Iterable[str]):
            return (p for p in paths if Path(p).exists())
        seen: Dict[str, None] = {}
        for process, paths in self.output_dict.items():
            if process not in _TERMINAL_PROCESSES:
                seen.update(dict.fromkeys(live(paths)))
        return list(seen)
