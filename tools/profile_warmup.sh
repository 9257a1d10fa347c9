cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wprof -- python3 tools/probe_warmup.py > /dev/null 2>&1
f=$(find /tmp/wprof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-170
