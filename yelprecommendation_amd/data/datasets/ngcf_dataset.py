"""reference data/datasets/ngcf_dataset.py:1-6 — the NGCF dataset IS the MF dataset."""
from .mf_dataset import MFDataset

NGCFDataset = MFDataset
