"""``ProjectFiles`` -- the unit of work handed between wrappers (reference util/data_classes.py:10-67):
``outputs/process/<input_stem>_<xxh64(file)[:8]>/source/<file>`` plus per-process output lists."""
import os
from shutil import copyfile
from typing import List, Union

import xxhash

from audiolab_amd.handlers import config


class ProjectFiles:
    def __init__(self, input_file):
        hash_gen = xxhash.xxh64()
        with open(input_file, "rb") as f:
            while chunk := f.read(8192):
                hash_gen.update(chunk)
        file_hash = hash_gen.hexdigest()[:8]
        project_name, _ = os.path.splitext(os.path.basename(input_file))
        project_dir = os.path.join(config.output_path, "process", f"{project_name}_{file_hash}")
        os.makedirs(project_dir, exist_ok=True)
        source_dir = os.path.join(project_dir, "source")
        os.makedirs(source_dir, exist_ok=True)
        src_file = os.path.join(source_dir, os.path.basename(input_file))
        if not os.path.exists(src_file):
            copyfile(input_file, src_file)
        self.src_file = src_file
        self.file_hash = file_hash
        self.project_dir = project_dir
        self.last_outputs = []
        self.video_sources = {}
        self.file_dict = {"source": [src_file]}
        self.output_dict = {}
        for root, _dirs, files in os.walk(project_dir):
            if root == project_dir:
                continue
            folder_name = os.path.basename(root)
            self.file_dict.setdefault(folder_name, [])
            for file in files:
                self.file_dict[folder_name].append(os.path.join(root, file))

    def add_output(self, process: str, outputs: Union[List[str], str]):
        if isinstance(outputs, str):
            outputs = [outputs]
        self.last_outputs = outputs
        self.file_dict.setdefault(process, [])
        self.output_dict.setdefault(process, [])
        self.file_dict[process].extend(outputs)
        self.output_dict[process].extend(outputs)

    def all_outputs(self) -> List[str]:
        output_list = []
        for key in self.output_dict:
            if key not in ("merge", "convert", "export"):
                for file in self.output_dict[key]:
                    if os.path.exists(file) and file not in output_list:
                        output_list.append(file)
        return output_list
