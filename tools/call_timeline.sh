#!/bin/bash
# Kernel / copy sequence of the last sample() call of tools/probe_call_timeline.py with start offsets and gaps (us).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=/tmp/ct_$$; rm -rf $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 tools/probe_call_timeline.py
python3 - "$O" <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') + ' ' + r.get('Size', r.get('Bytes', ''))))
ev.sort()
last = ev[-40:]
t0 = last[0][0]; prev = None
for s, e, n in last:
    print('%9.1f  dur %8.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0, n))
    prev = e
PY
