"""CPU oracle for the MObI camera+lidar sampling path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU (fp32) / numpy
(int, fp64) restatement of the reference algorithm for the hot path named in
BASELINE.json (`DDIMSampler`/`PLMSSampler` -> `LatentDiffusion.apply_model` ->
`UNetModel` + `AutoencoderKL` encode/decode).  Each function cites the reference
file:line it follows (paths relative to the reference repo root).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker.  Nothing under `mobi_amd/`
imports it; the product path raises when the HIP library is missing.

Parity pin: the reference has no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference's
own Python modules, imported on CPU in the build container by
`tests/golden/make_golden.py`; those outputs are committed under
`tests/golden/*.npz` and `tests/test_oracle_golden.py` checks the oracle
against every one of them.
"""
