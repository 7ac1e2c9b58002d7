#!/bin/bash
# Read-only probe of the GPU box's partition state: how many logical devices does the lease show,
# which compute / memory partition modes does the driver offer, and is the sysfs knob writable from this container?
out=${1:-gpurun_out/partition_probe.txt}
{
echo "== date"; date -u
echo "== devices seen by HIP"; python3 - <<'PY'
import torch
print("device_count", torch.cuda.device_count())
for i in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(i)
    print(i, p.name, p.multi_processor_count, "CUs", round(p.total_memory / 2**30, 1), "GiB", getattr(p, "gcnArchName", ""))
PY
echo "== rocm-smi --showcomputepartition --showmemorypartition"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -40
echo "== amd-smi partition"; (amd-smi partition 2>&1 || true) | head -60
echo "== sysfs knobs"
for d in /sys/class/drm/card*/device; do
  [ -e "$d/current_compute_partition" ] || continue
  echo "$d: current=$(cat $d/current_compute_partition 2>&1) available=$(cat $d/available_compute_partition 2>&1) mem=$(cat $d/current_memory_partition 2>&1)"
  if [ -w "$d/current_compute_partition" ]; then echo "  writable: yes (by mode bits)"; else echo "  writable: no"; fi
done
echo "== /dev/dri, /dev/kfd"; ls -la /dev/dri /dev/kfd 2>&1
echo "== mount of /sys"; grep -E ' /sys(/|\s)' /proc/mounts | head
echo "== kfd topology nodes"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n simd_count=$(grep -s simd_count $n/properties | awk '{print $2}') gfx=$(grep -s gfx_target_version $n/properties | awk '{print $2}')"; done
} > "$out" 2>&1
echo "probe written to $out"
