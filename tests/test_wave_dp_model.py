"""CPU tier: the lane-parallel DP formulation the HIP kernel implements (tests/wave_dp_model.py) is exact with
respect to the oracle on realistic gap-fill and X-drop extension problems, including tie-heavy scoring."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import oracle
import yaha_amd as ya
from wave_dp_model import wave_dp, supported
from problems import dp_problems_from_chain, batch_arrays, COMP


@pytest.mark.parametrize("reads,extra", [("r1k.fa", []), ("rchim.fa", ["-GOC", "0", "-GEC", "1"]), ("rq.fq", ["-BW", "7", "-X", "40"])])
def test_lane_model_equals_oracle(work, index11, reads, extra):
    with ya.Session(["-x", index11, "-q", os.path.join(work, reads)] + extra) as s:
        b = s.next_batch(60)
        probs = dp_problems_from_chain(s, b, limit=350, seed=3)
        exp = oracle.dp_batch(s.index, s.params, b, probs)
        bases, offs, codes = batch_arrays(s, b)
        n = 0
        for p, e in zip(probs, exp):
            if not supported(s.params, p.mode, p.qLen, p.rLen):
                continue
            f = codes[offs[p.read]:offs[p.read + 1]]
            q = f if not p.strand else np.array([COMP[c] for c in f[::-1]], dtype=np.uint8)
            assert wave_dp(s.params, p.mode, q, p.qOff, p.qLen, bases, s.index.maxROff, p.rOff, p.rLen) == e
            n += 1
        assert n > 200
