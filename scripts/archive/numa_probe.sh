#!/bin/bash
# Where do the host threads and the pinned staging sit relative to the GPU?  Prints the NUMA layout of the box and
# runs the 4 x 16-worker config-4 leg (scripts/host_batch64_rate.py, default staging) unbound and bound (taskset) to
# each NUMA node's CPUs.  Needs a GPU; taskset from util-linux.
cd "$GRAFT_REPO_ROOT"
echo "== nodes"; for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4,$5}')"; done
echo "== gpu"; for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "$d -> $(readlink -f $d | xargs basename) numa_node $(cat $d/numa_node) vendor $(cat $d/vendor)"; done
echo "== affinity of this shell: $(taskset -p $$ 2>/dev/null)"; nproc
export ONLY_DEFAULT=1 REPS=${REPS:-500}
run() { echo "== $1"; shift; "$@" python scripts/host_batch64_rate.py 2>&1 >/dev/null | grep -v amdgpu | tail -1; }
run unbound env
for n in /sys/devices/system/node/node*; do
  cpus=$(cat $n/cpulist)
  run "bound to $(basename $n) ($cpus)" taskset -c $cpus
done
