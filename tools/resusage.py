"""Tabulates hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin): one line per kernel.
    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2>&1 | python tools/resusage.py"""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin.read().splitlines():
    m = re.search(r"remark: ([A-Za-z \[\]/]+?): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    g = lambda k: str(r.get(k, "?"))
    print(f'{name[:120]:120s} V {g("VGPRs"):>4} A {g("AGPRs"):>3} S {g("SGPRs"):>3} scratch {g("ScratchSize [bytes/lane]"):>4} occ {g("Occupancy [waves/SIMD]")}')
