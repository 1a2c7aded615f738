#!/bin/bash
# gpurun_out/round (tools/collect_round.sh on the GPU box) -> profiles/rNN_* (tracked).  usage: tools/copy_round.sh 05
set -e
cd "$(dirname "$0")/.."
N=${1:?round number, two digits}
R=gpurun_out/round; P=profiles/r${N}
cp $R/bench_ssg.json ${P}_bench_line.json
cat $R/bench_pointnet2_cls_bga.json $R/bench_pointnet2_cls_msg.json $R/bench_dgcnn.json $R/bench_dgcnn_bga.json | grep '^{' > ${P}_bench_lines_other_models.json
cat $R/bench_ssg_det.json $R/bench_dgcnn_det.json | grep '^{' > ${P}_bench_deterministic.json
cp $R/kt_ssg/p_kernel_stats.csv ${P}_bench_kernel_stats.csv; cp $R/kt_ssg_by_grid.csv ${P}_bench_kernel_stats_by_grid.csv
cp $R/kt_dgcnn/p_kernel_stats.csv ${P}_dgcnn_kernel_stats.csv; cp $R/kt_dgcnn_by_grid.csv ${P}_dgcnn_kernel_stats_by_grid.csv
cp $R/pmc_traffic.json ${P}_pmc_traffic.json; cp $R/pmc_traffic_detail.json ${P}_pmc_traffic_detail.json
cp $R/pmc_traffic_detail_dgcnn.json ${P}_pmc_traffic_detail_dgcnn.json
cp $R/pmc_traffic_detail_pointnet2_cls_msg.json ${P}_pmc_traffic_detail_msg.json
cp $R/pmc_mfma.json ${P}_pmc_mfma.json
cp $R/pmc_insts_ssg.txt ${P}_pmc_insts.txt; cp $R/pmc_insts_dgcnn.txt ${P}_pmc_insts_dgcnn.txt
{ echo "# tools/bench_edgeconv.py (cfg3 graph, real kNN lists) -- entry-point times, per-kernel split (rocprofv3 --kernel-trace), PMC FETCH x2 / WRITE"
  grep -v amdgpu.ids $R/plain.txt; echo; cat $R/kernels.txt; echo; cat $R/traffic.txt; } > ${P}_edgeconv_micro.txt
grep -v amdgpu.ids $R/knn_bench.txt > ${P}_knn_bench.txt
[ -f $R/launches_dgcnn.txt ] && cp $R/launches_dgcnn.txt ${P}_dgcnn_launches.txt
[ -f $R/launches_pointnet2_cls_ssg.txt ] && cp $R/launches_pointnet2_cls_ssg.txt ${P}_ssg_launches.txt
python tools/traffic_table.py ${P}_pmc_traffic_detail_dgcnn.json dgcnn > ${P}_pmc_traffic_dgcnn.json
python tools/traffic_table.py ${P}_pmc_traffic_detail_msg.json msg > ${P}_pmc_traffic_msg.json
if [ -f $R/pmc_traffic_detail_pointnet2_cls_bga.json ]; then
  cp $R/pmc_traffic_detail_pointnet2_cls_bga.json ${P}_pmc_traffic_detail_bga.json
  python tools/traffic_table.py ${P}_pmc_traffic_detail_bga.json bga > ${P}_pmc_traffic_bga.json
fi
[ -f $R/pmc_dgrad_bf3.txt ] && cp $R/pmc_dgrad_bf3.txt ${P}_pmc_dgrad_bf3.txt
[ -f $R/pmc_dgrad_f32.txt ] && cp $R/pmc_dgrad_f32.txt ${P}_pmc_dgrad_f32.txt
[ -f $R/bwd_fused_gw_micro.txt ] && grep -v amdgpu.ids $R/bwd_fused_gw_micro.txt > ${P}_bwd_fused_gw_micro.txt
ls -la profiles | grep "r${N}_" | wc -l
