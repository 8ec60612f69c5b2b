timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do echo "dense: $(timeout 200 python bench.py --steps 20 --warmup 3 --cpu-blocks 0 2>/dev/null | python tools/benchline.py)"; done
