"""Minimal sequential-PHYLIP reader for the test fixtures (tests/golden/example.phy is the data file
the reference's own test scripts use, test_scripts/test_configs.txt).  DNA only; state encoding as
alignment.cpp:470-472 / Alignment::convertState: A,C,G,T = 0..3, IUPAC ambiguity = 4..17 (bitmask+3),
gap / ? / N = 18."""
import numpy as np

_DNA = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3, "-": 18, "?": 18, ".": 18, "N": 18, "X": 18,
        "R": 1 + 4 + 3, "Y": 2 + 8 + 3, "W": 1 + 8 + 3, "S": 2 + 4 + 3, "M": 1 + 2 + 3, "K": 4 + 8 + 3,
        "B": 2 + 4 + 8 + 3, "H": 1 + 2 + 8 + 3, "D": 1 + 4 + 8 + 3, "V": 1 + 2 + 4 + 3}


def read_phylip_dna(path):
    with open(path) as f:
        ntax, nsite = [int(x) for x in f.readline().split()[:2]]
        names, rows = [], []
        for _ in range(ntax):
            parts = f.readline().split()
            names.append(parts[0])
            seq = "".join(parts[1:]).upper()
            assert len(seq) == nsite, (parts[0], len(seq), nsite)
            rows.append([_DNA[ch] for ch in seq])
    return names, np.array(rows, dtype=np.uint8)
