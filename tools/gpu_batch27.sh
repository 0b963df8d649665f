#!/bin/bash
# round-2 batch 27: where does an 8-rank share of the headline frame lose its 10 %?
set -o pipefail
O=gpurun_out/r3d; mkdir -p $O
for spp in 2500 1250; do RANKS=1,8 timeout -k 10 300 python tools/partition_balance.py masterchief $spp 2>/dev/null | cut -c1-330 | tee -a $O/shares.txt; done
RANKS=8 QUEUES=1,4,16,64 timeout -k 10 300 python tools/partition_balance.py masterchief 5000 2>/dev/null | cut -c1-330 | tee -a $O/shares.txt
RANKS=8 UNITS=1,2,4,16 timeout -k 10 300 python tools/partition_balance.py masterchief 5000 2>/dev/null | cut -c1-330 | tee -a $O/shares.txt
RANKS=8 CHUNKS=314 timeout -k 10 300 python tools/partition_balance.py masterchief 5000 2>/dev/null | cut -c1-330 | tee -a $O/shares.txt
RANKS=8 CHUNKS=628 timeout -k 10 300 python tools/partition_balance.py masterchief 5000 2>/dev/null | cut -c1-330 | tee -a $O/shares.txt
