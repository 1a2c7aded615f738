#!/bin/bash
# Runs ON THE GPU BOX: what the box says about itself next to one short SSG bench line -- the boxes of the pool differ by up to 7 %
# on the same tree and neither the shader clock nor the tree explains it.  usage: tools/box_probe.sh [tag]
cd "$(dirname "$0")/.."
T=${1:-x}; O=gpurun_out/box_$T.txt
{
  echo "== host"; uname -r; nproc; grep -m1 "model name" /proc/cpuinfo
  echo "== rocm-smi"; rocm-smi --showproductname --showclocks --showmaxpower --showpower --showperflevel --showvoltage --showtemp 2>&1 | grep -v "^$" | head -60
  echo "== hwmon"; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_max /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/pp_dpm_fclk /sys/class/drm/card*/device/pp_dpm_sclk; do [ -r $f ] && { echo $f; cat $f; }; done
  echo "== this process's GPU"
  python -c "
import torch, os, glob
pr = torch.cuda.get_device_properties(0)
addr = '%04x:%02x:%02x.0' % (int(getattr(pr, 'pci_domain_id', 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
print(pr.name, addr, [c for c in glob.glob('/sys/class/drm/card*') if os.path.basename(os.path.realpath(c + '/device')) == addr])
print('GPUs of the node and their power right now (W):', {os.path.basename(os.path.realpath(f.split('/hwmon/')[0])): int(open(f).read()) / 1e6 for f in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average')})
" 2>&1 | grep -v amdgpu.ids
  echo "== bench (SSG, 20 steps)"
  python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print('%.1f clouds/s  %.3f ms/step  dominant %s %.1f us  clock %s' % (d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], d.get('clock')))
for k in d['kernels'][:8]:
    print('   %-28s %-40s %8.1f us' % (k['kernel'], k['shape'], k['avg_us']))
"
  echo "== rocm-smi under load is in the bench line's clock object (hwmon poller)"
} > $O 2>&1
cat $O
