#!/bin/bash
# round 5: the look-ahead factorisation of k_sample_wg2 (bpmf_amd/csrc/variants/la.so: BPMF_PATCH=tools/patches/wg2_lookahead.patch bash tools/build_variant.sh la tools/patches/apply.py) against the tree:
# bit-identity of a short chain, the K = 128 parity tests, interleaved bench of both K = 128 workloads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cat > /tmp/chain_sha.py <<'PY'
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, bpmf_amd
from bpmf_amd import synth
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42)
for dt in ("f32", "f64"):
    eng = bpmf_amd.HipEngine(128, dtype=dt)
    res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=3, burnin=1, Tt=Tt, pipelined=True)
    print(dt, hashlib.sha1(np.ascontiguousarray(res["U"]).tobytes()).hexdigest()[:16], hashlib.sha1(np.ascontiguousarray(res["V"]).tobytes()).hexdigest()[:16], "%.9f" % res["rmse"][-1])
    eng.close()
PY
echo "== chain sha, tree then variant"
timeout 300 python /tmp/chain_sha.py 2>&1 | tail -3
BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/la.so timeout 300 python /tmp/chain_sha.py 2>&1 | tail -3
echo "== parity tests K = 128"
BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/la.so timeout 900 python -m pytest tests -m gpu -x -q -k "128 or f32 or fp32" 2>&1 | tail -5
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()}, 'frac %.3f' % r['frac'])"; }
for r in 1 2; do
  for wl in ml1m_k128 ml1m_k128_f64; do
    for lib in tree la; do
      E=""; [ $lib = la ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/la.so"
      env $E timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 60 --warmup 10 2>/dev/null | line "$wl $lib"
    done
  done
done
