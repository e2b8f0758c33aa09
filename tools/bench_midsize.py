"""The iteration rate of config 3 at reduced linear scales, over a long region WITHOUT profiling events (so that SKF_GRAPH=1,
the hipGraph replay of the iteration, is honoured):   python tools/bench_midsize.py [scale ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    for sc in [float(a) for a in sys.argv[1:]] or [0.05, 0.1, 0.2, 0.3]:
        w = bench.run_workload('c3', os.environ.get('SKF_BENCH_DTYPE', 'bf16'), 20, 3, scale=sc, sustained=1000)
        print('scale %.2f: %.1f it/s in the 20-step window, %.1f it/s over 1000 steps (%d launches per iteration, host enqueue %.2f ms)'
              % (sc, 20 / w['elapsed'], w['sustained']['value'], w['launches_per_step'], w['enqueue_ms_per_step']), flush=True)


if __name__ == '__main__':
    main()
