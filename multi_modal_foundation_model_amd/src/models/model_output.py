"""Base output type (reference: models/model_output.py:11-17)."""
from dataclasses import dataclass, fields
from typing import Optional

import torch


@dataclass
class ModelOutput:
    loss: Optional[torch.FloatTensor] = None
    n_examples: Optional[torch.LongTensor] = None

    def to_dict(self):
        return {f.name: getattr(self, f.name) for f in fields(self)}
