#!/bin/bash
# tools/kernel_resources.sh UNIT [extra hipcc flags] : registers / scratch / occupancy of every kernel of one translation unit
# (csrc/device/UNIT.hip), from the compiler's own resource remarks (no GPU needed)
cd "$(dirname "$0")/.."
u=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc -Wno-unused-function -I include "$@" \
  -Rpass-analysis=kernel-resource-usage -c rabitq-rs_amd/csrc/device/$u.hip -o /tmp/kr_$u.o 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
def flush():
    if cur:
        n = subprocess.run(['c++filt', cur['Function Name']], capture_output=True, text=True).stdout.strip()
        n = re.sub(r'\(.*', '', n).replace('void rbq::', '')
        print('%-34s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %2s lds %6s' % (n, cur.get('VGPRs'), cur.get('AGPRs'), cur.get('TotalSGPRs'), cur.get('ScratchSize [bytes/lane]'), cur.get('Occupancy [waves/SIMD]'), cur.get('LDS Size [bytes/block]')))
for line in sys.stdin:
    m = re.search(r'remark: [^:]*:\d+:\d+:\s+(.*?): (\S+) \[-Rpass', line) or re.search(r'remark:\s+(.*?): (\S+) \[-Rpass', line)
    if not m:
        if 'error' in line or 'warning' in line: sys.stdout.write(line)
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name': flush(); cur = {}
    cur[k] = v
flush()
"
