"""Test-side alias of the package's model builders (multi_modal_foundation_model_amd/builders.py)."""
import conftest  # noqa: F401  (puts the repo root and the API mirror on sys.path)
from multi_modal_foundation_model_amd.builders import (SRC, build_model, build_model_mods, load_config, make_optimizer, model_config,  # noqa: F401
                                                        tiny_config)
