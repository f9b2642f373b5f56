#!/bin/bash
# the whole GPU suite with its parity report, then smoke()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite.log 2>&1; rc=$?
tail -3 gpurun_out/suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
