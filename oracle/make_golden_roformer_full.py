"""TEST INFRASTRUCTURE.  Output of THIS repo's Roformer oracle (oracle/roformer_oracle.py: unpinned restatement) at the reference's full
model sizes -- Mel-Band RoFormer depth 6 and BS-RoFormer depth 12, dim 384, one 8 s chunk of the synthetic mix -- in float32 and in the
half-precision storage mode, cached so that the GPU suite can assert the FULL depth without spending ten CPU minutes per run
(the depth-12 model had no full-depth parity run).  Every 7th sample of both channels is kept (1.6 MB instead of 11).

    python oracle/make_golden_roformer_full.py      -> tests/golden/roformer_full.npz      (about 8 min on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiolab_amd.synth import synth_mix      # noqa: E402  (deterministic test signal: data, not product code under test)
from oracle import roformer_oracle as ro      # noqa: E402

STRIDE = 7
FULL_DEPTH = {"mel": 6, "bs": 12}


def main():
    out = {"stride": np.int64(STRIDE)}
    for kind, depth in FULL_DEPTH.items():
        cfg = ro.RoformerConfig(kind=kind, depth=depth)
        sd = ro.synthetic_state_dict(cfg, 0)
        x = torch.from_numpy(synth_mix(cfg.chunk_size))
        for half in (False, True):
            t0 = time.time()
            with torch.no_grad():
                y = ro.forward(cfg, sd, x[None], half=half)[0].numpy()
            key = f"{kind}_{'half' if half else 'f32'}"
            out[key] = y[..., ::STRIDE].astype(np.float32)
            out[key + "_peak"] = np.float32(np.max(np.abs(y)))
            print(key, y.shape, f"{time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "roformer_full.npz"), **out)


if __name__ == "__main__":
    main()
