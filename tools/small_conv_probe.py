#!/usr/bin/env python
"""Hot re-issue timing of the <= 4-channel conv launches (the streaming VALU kernels of conv.hip) at the shapes of the
bench workloads: FashionMNIST's Conv2d(1, 64) / ConvTranspose2d(64, 1), CelebA's Conv2d(3, 32) / ConvTranspose2d(32, 3).
    python tools/small_conv_probe.py            (MVAE_HIP_LIB selects the build)"""
import sys

from gemm_bench import conv_cases, convT_cases, timeit

CASES = (conv_cases(1024, 1, 28, 64, 2, 1, 'fashion enc1 1->64 28x28 B1024')
         + convT_cases(2048, 64, 14, 1, 2, 1, 'fashion dec3 64->1 14x14 B2048')
         + conv_cases(256, 3, 64, 32, 2, 1, 'celeba enc1 3->32 64x64 B256')
         + convT_cases(512, 32, 32, 3, 2, 1, 'celeba dec4 32->3 32x32 B512')
         + convT_cases(256, 32, 32, 3, 2, 1, 'celeba19 dec4 32->3 32x32 B256'))


def main():
    print('%-48s %8s %8s' % ('op', 'us', 'GFLOP'))
    for name, fl, fn in CASES:
        ms = timeit(fn)
        print('%-48s %8.1f %8.2f' % (name, ms * 1e3, fl / 1e9))
    sys.stdout.flush()


if __name__ == '__main__':
    main()
