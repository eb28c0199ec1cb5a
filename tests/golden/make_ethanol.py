"""Data fixture for BASELINE config 1 (SOAP-BPNN on the bundled ethanol data set): the first frames of the reference's
``tests/resources/ethanol_reduced_100.xyz`` (9-atom molecules, energies and forces) as arrays.

  python tests/golden/make_ethanol.py        # writes tests/golden/ethanol_first10.npz (runs in the build container only)

Inputs and labels only (positions, atomic numbers, energies, forces): data the reference's own tests hold, no code.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tests/resources/ethanol_reduced_100.xyz"
NUMBERS = {"H": 1, "C": 6, "O": 8}


def read_frames(path, n_frames):
    frames = []
    with open(path) as fh:
        lines = fh.read().splitlines()
    k = 0
    while k < len(lines) and len(frames) < n_frames:
        n = int(lines[k])
        energy = float(lines[k + 1].split("energy=")[1].split()[0])
        rows = [ln.split() for ln in lines[k + 2:k + 2 + n]]
        z = np.array([NUMBERS[r[0]] for r in rows], dtype=np.int32)
        pos = np.array([[float(x) for x in r[1:4]] for r in rows])
        forces = np.array([[float(x) for x in r[4:7]] for r in rows])
        frames.append((z, pos, forces, energy))
        k += 2 + n
    return frames


if __name__ == "__main__":
    frames = read_frames(SRC, 10)
    np.savez(os.path.join(HERE, "ethanol_first10.npz"),
             species=np.stack([f[0] for f in frames]), positions=np.stack([f[1] for f in frames]),
             forces=np.stack([f[2] for f in frames]), energies=np.array([f[3] for f in frames]))
    print("frames", len(frames), "atoms", frames[0][0].shape)
