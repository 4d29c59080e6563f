"""Paths (reference handlers/config.py:1-6).  ``output_path`` is where ProjectFiles keeps
``process/<name>_<hash8>/``; override with AUDIOLAB_OUTPUT_PATH when embedding."""
import os

app_path = os.environ.get("AUDIOLAB_APP_PATH", os.getcwd())
output_path = os.environ.get("AUDIOLAB_OUTPUT_PATH", os.path.join(app_path, "outputs"))
model_path = os.environ.get("AUDIOLAB_MODEL_PATH", os.path.join(app_path, "models"))
