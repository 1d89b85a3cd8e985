cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | tail -15
python bench.py 2>&1 | tail -1
python bench.py --scenes 8 2>&1 | tail -1
python bench.py --grid 64 2>&1 | tail -1
