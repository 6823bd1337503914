"""Configuration / logging / seeding entry points.

Counterpart of the reference's ``EEG_CODE/config.py`` (Config :19-80,
setup_logging :83-94, set_seed :97-105): same attribute names, defaults and YAML
overlay rule (only existing attributes are overwritten).  Two aliases the
reference's own ``run_training_lite.main()`` reads but ``Config`` never defined
(``config.freq_bands`` run_training_lite.py:372, ``config.epochs`` :467) are
provided as properties so the entry point actually runs.
Added for the MI355X trainer: ``synthetic`` data shape block.
"""
from __future__ import annotations

import logging
import os
import random
from pathlib import Path
from typing import Optional

import numpy as np
import torch
import yaml


class Config:
    def __init__(self, config_path: Optional[str] = None, make_dirs: bool = True):
        root = Path(os.getenv("EEG_DATA_PATH", r"E:\Head_neck"))
        cleaned = root / "EEG" / "DATA" / "PROC" / "data_proc" / "cleaned_data"
        self.base_path = root
        self.eeg_path_pw = cleaned / "TF_dir" / "pwspctrm" / "PWS" / "feat"
        self.eeg_path_erp = cleaned / "TF_dir" / "ERP" / "New"
        self.eeg_path_conn = cleaned / "conn_dir" / "CONN"
        self.label_path = cleaned / "TF_dir"

        self.subject_list = list(range(1, 64))
        self.bands = {"alpha": "Alpha", "beta": "Beta", "theta": "Theta"}
        self.eeg_segments = [f"{f}_Hz" for f in (1, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 25, 30, 40)]
        self.func_segments = ["open", "close"]

        self.batch_size = 8
        self.num_epochs = 50
        self.learning_rate = 5e-5
        self.weight_decay = 1e-5
        self.patience = 10
        self.n_splits = 5
        self.grad_clip = 1.0

        self.fusion_dim = 128
        self.hidden_dim = 64
        self.dropout = 0.65

        self.output_dir = Path("./results")
        self.log_dir = Path("./logs")
        self.checkpoint_dir = Path("./checkpoints")

        # synthetic stand-in for the private clinical data (BASELINE config #1)
        self.synthetic = {"subjects": 40, "erp_channels": 8, "pw_channels": 8,
                          "samples": 256, "conn_features": 459, "seed": 1234}

        if make_dirs:
            for d in (self.output_dir, self.log_dir, self.checkpoint_dir):
                Path(d).mkdir(parents=True, exist_ok=True)
        if config_path and os.path.exists(config_path):
            self.load_config(config_path)

    # aliases read by run_training_lite.main() in the reference
    @property
    def freq_bands(self):
        return self.eeg_segments

    @property
    def epochs(self):
        return self.num_epochs

    def load_config(self, path: str):
        with open(path, "r") as fh:
            for key, value in (yaml.safe_load(fh) or {}).items():
                if key in self.__dict__:
                    setattr(self, key, value)

    def save_config(self, path: str):
        dump = {k: (str(v) if isinstance(v, Path) else v)
                for k, v in self.__dict__.items() if not k.startswith("_")}
        with open(path, "w") as fh:
            yaml.dump(dump, fh, default_flow_style=False)


def setup_logging(log_dir: Path, name: str = "eeg_analysis"):
    Path(log_dir).mkdir(parents=True, exist_ok=True)
    logging.basicConfig(
        level=logging.INFO,
        format="%(asctime)s - %(name)s - %(levelname)s - %(message)s",
        handlers=[logging.FileHandler(Path(log_dir) / f"{name}.log"), logging.StreamHandler()])
    return logging.getLogger(name)


def set_seed(seed: int = 42):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
