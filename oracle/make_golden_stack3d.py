"""Generate tests/golden/stack3d.npz: the reference's 3-D input mode of ``log_space_fft_filtering``
(``filtering.py:182-183, 210-211``: one ``pywt.wavedec2`` over the last two axes, ONE Otsu threshold per level for
the whole stack, row medians / FFT per plane row) run by the REAL reference in this container.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/code \
        /opt/conda/bin/python3.9 -W ignore oracle/make_golden_stack3d.py

Only inputs / outputs (data) are written.  Test infrastructure: nothing in the product imports this.
"""

import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(REPO, "tests", "golden")

from aind_smartspim_destripe import filtering as ref  # noqa: E402  (the reference itself)

_spec = importlib.util.spec_from_file_location("dsx_synth", os.path.join(REPO, "aind_smartspim_destripe_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def main():
    import pywt
    import scipy
    import skimage

    d = {"versions": json.dumps({"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                                 "pywt": pywt.__version__, "skimage": skimage.__version__,
                                 "reference": "AllenNeuralDynamics/aind-smartspim-destripe @ 2025-05-23 (/root/reference)"})}
    cases = []
    for name, (n, h, w) in (("a", (5, 64, 96)), ("b", (3, 101, 80))):
        stack = np.stack([synth.synthetic_plane(k, h, w) for k in range(n)])
        stack[0, 10:20, 30:40] += 3000  # one plane with a bright block: its coefficients set the stack's histogram range
        d[name + "__in"] = stack
        for cfg_name, cfg in (("cells", synth.CELLS_CONFIG), ("nocells", synth.NO_CELLS_CONFIG)):
            for lvl in (None, 2):
                for dt in ("u16", "f32"):
                    x = stack if dt == "u16" else stack.astype(np.float32)
                    out = ref.log_space_fft_filtering(x, wavelet=cfg["wavelet"], level=lvl, sigma=cfg["sigma"],
                                                      max_threshold=cfg["max_threshold"])
                    case = "{}__{}__{}__{}".format(name, cfg_name, "Lmax" if lvl is None else "L%d" % lvl, dt)
                    d[case + "__out"] = np.asarray(out)
                    cases.append(case)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(GOLDEN, "stack3d.npz"), **d)
    print("stack3d:", len(cases), "cases")


if __name__ == "__main__":
    main()
