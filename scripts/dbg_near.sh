for i in $(seq 1 24); do python bench.py --near-arm --steps 2 --warmup 1 --min-seconds 0 --cpu-seconds 0 --check-frames 1 --overlap-pipelines 0 2>&1 | tail -1 | cut -c1-100; done
