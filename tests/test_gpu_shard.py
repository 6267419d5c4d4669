"""BASELINE.json configs[3] -- "Synthetic R 10M users x 1M items, 200 nnz/row, K=32, 8 x MI355X sharded": one rank's
share of the REAL matrix (bpmf_amd/synth_dev.py: device-generated, the same matrix whatever the rank count; rank 7 of 8
= user chunk 7 and the item range with the most columns), both half-iterations on one GPU with random factors:

  * spot-checked against the oracle: the heaviest, the lightest and random columns of both sides (a column's ratings
    are pulled off the device, the rows it reads gathered into a small factor; oracle.sample_column keys the RNG
    stream on the GLOBAL column id), 1e-9 of max|U| as everywhere;
  * size-independent properties of the whole shard: the returned sums are the sums of the sampled columns
    (checksum of checksums), bit-reproducible when run twice, columns outside the rank's range untouched.
The factors of this configuration live in HBM (U: 2.56 GB), not in L2 -- the regime the ML-1M tests cannot reach.
"""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_one_rank_of_the_10Mx1M_configuration():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py")], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0 and "SHARD-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    print(r.stdout[-300:])
