set -x
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 2>&1 | tail -3
