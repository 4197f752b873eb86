#!/bin/bash
# development: does the prover's host thread care which socket it runs on?  The same bench lines with the process bound
# to the CPUs local to the GPU (sysfs local_cpulist of the visible device), to the other socket's, and unbound.
cd "${GRAFT_REPO_ROOT:-.}"
bdf=$(python3 - <<'P'
import ctypes
h = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
assert h.hipDeviceGetPCIBusId(buf, 64, 0) == 0
print(buf.value.decode().lower())
P
)
loc=$(cat /sys/bus/pci/devices/$bdf/local_cpulist)
node=$(cat /sys/bus/pci/devices/$bdf/numa_node)
echo "device $bdf numa_node $node local_cpulist $loc"
all=$(cat /sys/devices/system/cpu/online)
other=$(python3 -c "
def parse(s):
    r=set()
    for p in s.split(','):
        a,_,b=p.partition('-'); r|=set(range(int(a),int(b or a)+1))
    return r
o=sorted(parse('$all')-parse('$loc'))
print(','.join(map(str,o)))")
for rep in 1 2; do
for cfg in "unbound:" "local:taskset -c $loc" "remote:taskset -c $other"; do
  name=${cfg%%:*}; pre=${cfg#*:}
  for size in 16 20; do
    $pre python bench.py --log-n $size --table range --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', $size, d['value'])"
  done
  $pre python bench.py --log-n 24 --table and --steps 10 --warmup 3 --no-cpu-baseline --no-inflight 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', 24, d['value'])"
done
done
