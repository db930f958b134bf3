cd $GRAFT_REPO_ROOT
for rep in 1 2; do for B in 0 1; do
echo -n "BALANCE=$B: "
GRAPHLILY_SPMV_BALANCE=$B timeout 300 python bench.py --steps 100 --no-bfs --no-cpu-baseline --no-pattern 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
