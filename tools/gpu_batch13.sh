#!/bin/bash
# round-2 batch 13: sibling-pair device node order vs pre-order (SRT_NODE_PAIRS), parity suite
set -o pipefail
O=gpurun_out/r2n; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; grep -a "passed\|failed\|Error\|^E  " $O/pytest.txt | tail -12
for rep in 1 2; do for np in 1 0; do
  SRT_NODE_PAIRS=$np timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-pmc > $O/head_$np_$rep.json 2>/dev/null
  echo "headline node_pairs $np rep $rep: $(python -c "import json;d=json.load(open('$O/head_$np_$rep.json'));print(d['value'], d['roofline']['kernel_ms_avg'])" 2>&1)"
done; done
for w in soup_1m_720p_16spp soup_10m_720p_16spp; do for np in 1 0; do
  SRT_NODE_PAIRS=$np timeout -k 10 900 python bench.py --workload $w --steps 3 --no-cpu-baseline > $O/${w}_$np.json 2>/dev/null
  echo "$w node_pairs $np: $(python -c "import json;d=json.load(open('$O/${w}_$np.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'], 'hbm', r.get('hbm_measured_GBps'), r.get('hbm_measured_frac'), 'alg', r.get('algorithmic_GBps'))" 2>&1)"
done; done
for w in iron_720p_5000spp spheres_720p_1024spp; do for np in 1 0; do
  SRT_NODE_PAIRS=$np timeout -k 10 300 python bench.py --workload $w --steps 3 --no-cpu-baseline --no-pmc > $O/${w}_$np.json 2>/dev/null
  echo "$w node_pairs $np: $(python -c "import json;d=json.load(open('$O/${w}_$np.json'));r=d['roofline'];print(d['value'], r['kernel_ms_avg'])" 2>&1)"
done; done
