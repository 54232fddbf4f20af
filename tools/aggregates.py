#!/usr/bin/env python
"""Aggregates of a per-(call, shape) table (profiles/rNN_by_shape.json): flops / rocprofv3 time over the conv launches, over
all GEMM-shaped launches, BatchNorm time -- the figures VERDICT r5 recomputed by hand.   python tools/aggregates.py [table.json]"""
import json
import sys

PEAK = 157.3


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r06_by_shape.json'
    t = json.load(open(path))
    print('# %s  (peak %.1f TFLOP/s fp32 MFMA)' % (path, PEAK))
    print('# workload      kernels us/step | conv GFLOP      us   TF/s  frac | all GEMM-shaped GFLOP   us   TF/s  frac | BatchNorm us | Adam us')
    for w in ('mnist', 'fashionmnist', 'celeba', 'celeba19'):
        conv_fl = conv_us = gemm_fl = gemm_us = bn_us = adam_us = tot = 0.0
        for k, v in t[w].items():
            calls = v.get('calls_per_step', 1.0)
            us = v['rocprof_avg_us'] * calls
            fl = (v.get('algorithmic_flops') or 0.0) * calls
            tot += us
            if k.startswith('bn_'):
                bn_us += us
            if k.startswith('adam'):
                adam_us += us
            if fl:
                gemm_fl += fl
                gemm_us += us
                if k.startswith('conv'):
                    conv_fl += fl
                    conv_us += us
        f = lambda fl, us: (fl / 1e9, us, fl / us / 1e6 if us else 0.0, fl / us / 1e6 / PEAK if us else 0.0)  # noqa: E731
        print('%-12s %12.1f | %10.2f %8.1f %6.1f %5.3f | %10.2f %8.1f %6.1f %5.3f | %8.1f | %6.1f'
              % ((w, tot) + f(conv_fl, conv_us) + f(gemm_fl, gemm_us) + (bn_us, adam_us)))


if __name__ == '__main__':
    main()
