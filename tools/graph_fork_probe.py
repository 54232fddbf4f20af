#!/usr/bin/env python
"""Probe: does hipGraph stream capture accept nested fork/join topologies (main -> side -> side2)
and repeated re-forks of the same stream?  Prints one line per pattern."""
import sys
import torch

dev = torch.device('cuda', 0)
a = torch.zeros(1 << 16, device=dev)


def run(pattern):
    s1, s2, s3 = (torch.cuda.Stream(dev) for _ in range(3))
    g = torch.cuda.CUDAGraph()
    bufs = [torch.zeros_like(a) for _ in range(4)]
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for rep in range(3):
            if pattern == 'flat':           # main forks s1 and s2, both join main
                for s, b in ((s1, bufs[0]), (s2, bufs[1])):
                    s.wait_stream(main)
                    with torch.cuda.stream(s):
                        b.add_(1.0)
                bufs[2].add_(1.0)
                main.wait_stream(s1); main.wait_stream(s2)
            elif pattern == 'nested':       # main forks s1; s1 forks s2; s2 joins s1; s1 joins main
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    bufs[0].add_(1.0)
                    s2.wait_stream(s1)
                    with torch.cuda.stream(s2):
                        bufs[1].add_(1.0)
                    bufs[3].add_(1.0)
                    s1.wait_stream(s2)
                bufs[2].add_(1.0)
                main.wait_stream(s1)
            elif pattern == 'nested_refork':  # like nested, s2 forked/joined several times inside
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    for k in range(3):
                        bufs[0].add_(1.0)
                        s2.wait_stream(s1)
                        with torch.cuda.stream(s2):
                            bufs[1].add_(1.0)
                    s1.wait_stream(s2)
                for k in range(3):
                    bufs[2].add_(1.0)
                    s3.wait_stream(main)
                    with torch.cuda.stream(s3):
                        bufs[3].add_(1.0)
                main.wait_stream(s3)
                main.wait_stream(s1)
            elif pattern == 'cross':        # flat forks, then s2 waits on s1 (sibling dependency)
                s1.wait_stream(main); s2.wait_stream(main)
                with torch.cuda.stream(s1):
                    bufs[0].add_(1.0)
                s2.wait_stream(s1)
                with torch.cuda.stream(s2):
                    bufs[1].add_(1.0)
                bufs[2].add_(1.0)
                main.wait_stream(s1); main.wait_stream(s2)
            elif pattern == 'nested_join_main':   # s2 forked from s1 but joined into main
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    bufs[0].add_(1.0)
                    s2.wait_stream(s1)
                    with torch.cuda.stream(s2):
                        bufs[1].add_(1.0)
                    bufs[3].add_(1.0)
                bufs[2].add_(1.0)
                main.wait_stream(s2); main.wait_stream(s1)
            elif pattern == 'sibling_join':  # flat forks; s1 waits on s2 (join into a forked stream); s1 joins main
                s1.wait_stream(main); s2.wait_stream(main)
                with torch.cuda.stream(s2):
                    bufs[1].add_(1.0)
                with torch.cuda.stream(s1):
                    bufs[0].add_(1.0)
                    s1.wait_stream(s2)
                    bufs[3].add_(1.0)
                bufs[2].add_(1.0)
                main.wait_stream(s1); main.wait_stream(s2)
            elif pattern == 'nested_refork_join_main':   # s2 re-forked from s1 repeatedly, joined into main only
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    for k in range(3):
                        bufs[0].add_(1.0)
                        s2.wait_stream(s1)
                        with torch.cuda.stream(s2):
                            bufs[1].add_(1.0)
                bufs[2].add_(1.0)
                main.wait_stream(s1); main.wait_stream(s2)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    print(pattern, 'ok', [float(b[0]) for b in bufs], flush=True)


run(sys.argv[1])
