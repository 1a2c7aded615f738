#!/bin/bash
# round 5, GPU call 1: prob_sample parity + PMC traffic of the cfg3 / cfg5 steps (baseline, before the kernel work)
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c1; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "prob_sample" > $O/pytest_prob.log 2>&1; echo "pytest rc=$?" >> $O/pytest_prob.log
timeout 600 python tools/collect_traffic.py --model dgcnn > $O/traffic_dgcnn.log 2>&1
timeout 600 python tools/collect_traffic.py --model pointnet2_cls_msg > $O/traffic_msg.log 2>&1
cp gpurun_out/pmc_traffic_detail_dgcnn.json gpurun_out/pmc_traffic_detail_pointnet2_cls_msg.json $O/ 2>/dev/null
tail -3 $O/pytest_prob.log
