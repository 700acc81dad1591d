"""BASELINE.json config 1 at its literal size (src/RUNME2_others_degrid_phantom.m:23-69): the 256^2 modified Shepp-Logan phantom
degridded on the CPU by the IRT NUFFT (contrib/irt restated in oracle/irt_nufft.py: nufft_init(2 pi traj, [256 256], [4 4], [512 512],
[128 128]), linear-radial trajectory of 512 readout x 512 spokes, theta = pi (pe - 1) / 512), written as the [1, 1, 512, 512, 1]
complex64 RawArray the reference's script writes (output/sl_data_irt.ra) -- and, when a GPU is there, the same phantom through
the HIP forward path (`tron sl.ra data.ra`, src/RUNME1_tron_degrid_phantom.sh:5) with RUNME2:96's figure beside it:
    Data NMSE = norm(irt - tron) / max|irt|
(IRT is an INDEPENDENT algorithm -- min-max interpolation in double precision -- so this is the paper's comparison, not parity.)

    python tools/config1.py [outdir]        writes sl.ra, sl_data_irt.ra and (GPU) sl_data_tron.ra, prints one JSON line
Test tooling: imports oracle/ (the IRT restatement), like tests/test_irt.py."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import irt_nufft as irt      # noqa: E402
from tron_amd import ra                   # noqa: E402


def irt_forward(N=256):
    """(phantom [N, N] as TRON lays it out, IRT samples [nro, npe]) for the trajectory of RUNME2:29-36."""
    nro = npe = 2 * N
    t0 = time.perf_counter()
    st = irt.Nufft(irt.radial_trajectory(nro, npe), (N, N), (4, 4), (2 * N, 2 * N), (N / 2, N / 2))
    x = irt.shepp_logan(N).T.copy()                              # first image index = x = TRON's column (cosine) axis
    X = st.forward(x).reshape((nro, npe), order="F")
    return x, X, time.perf_counter() - t0


def data_nmse(irt_data, tron_data):
    """RUNME2:96."""
    return float(np.linalg.norm((irt_data - tron_data).ravel()) / np.abs(irt_data).max())


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "."
    os.makedirs(out, exist_ok=True)
    N = 256
    x, X, secs = irt_forward(N)
    img = np.zeros((1, 1, N, N, 1), np.complex64, order="F")
    img[0, 0, :, :, 0] = x
    ra.write(os.path.join(out, "sl.ra"), img)
    ra.write(os.path.join(out, "sl_data_irt.ra"), np.asfortranarray(X.astype(np.complex64)[None, None, :, :, None]))
    line = {"config": "BASELINE.json configs[0]: Shepp-Logan 256^2 -> 512 readout x 512 linear-radial spokes, contrib/irt restated (CPU, double)",
            "irt_seconds": round(secs, 2), "irt_max_abs": float(np.abs(X).max()), "files": ["sl.ra", "sl_data_irt.ra"]}
    from tron_amd import lib
    if lib.device_count() >= 1:
        got, dims = lib.recon(img, adjoint=False)                # tron sl.ra data.ra: linear angles, u = 1 -> 512 x 512
        T = got[0, 0, :, :, 0].astype(np.complex128)
        ra.write(os.path.join(out, "sl_data_tron.ra"), got)
        s = np.vdot(X, T) / np.vdot(X, X)                        # (the two transforms agree up to a constant: RUNME2 normalises both)
        line.update(data_nmse=data_nmse(X * s, T), scale=[float(s.real), float(s.imag)],
                    max_abs_magnitude_difference=float(np.abs(np.abs(T) - np.abs(X * s)).max() / np.abs(X).max()),
                    files=line["files"] + ["sl_data_tron.ra"])
    print(json.dumps(line))


if __name__ == "__main__":
    main()
