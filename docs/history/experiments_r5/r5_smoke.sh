set -u
O=gpurun_out/r5smoke; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
