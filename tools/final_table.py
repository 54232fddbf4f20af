#!/usr/bin/env python
"""Rows of DESIGN.md section 7's final table from the committed collection (profiles/<pfx>_bench_default.json + <pfx>_by_shape.json):
    python tools/final_table.py [r06]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pfx = sys.argv[1] if len(sys.argv) > 1 else 'r06'
d = json.loads([l for l in open(os.path.join(ROOT, 'profiles', pfx + '_bench_default.json')) if l.startswith('{')][-1])
r = d['roofline']
print('bench line: %.4f ms/step (median %.4f), %.0f images/s; %s in situ %.4f rocprof %.4f (%.2f us) hot %.4f; traffic %.2f MB; cpu %.0f' % (
    d['ms_per_step'], d['ms_per_step_median'], d['value'], r['kernel'], r['frac'], r['rocprof_frac'], r['rocprof_avg_us'],
    r['hot_cache_reissue']['frac'], r['traffic'] / 1e6, d['cpu_baseline']['value']))
print('  in-situ interval %.2f us, achieved %.1f TFLOP/s; elbo_delta %s; capture_step %.3f ms' % (
    r['avg_launch_ms'] * 1e3, r['achieved'], d['cpu_baseline']['elbo_delta'], d['module_surface']['FusedAdam + capture_step']['ms_per_step']))
for a in d['also']:
    ra = a['roofline']
    print('%-14s %.3f ms/step, %.0f images/s; %s in situ %.4f rocprof %.4f hot %.4f; cpu %.1f; capture_step %s' % (
        a['workload'].split()[0], a['ms_per_step'], a['value'], ra['kernel'], ra['frac'], ra['rocprof_frac'],
        ra['hot_cache_reissue']['frac'], a['cpu_baseline']['value'],
        a.get('module_surface', {}).get('FusedAdam + capture_step', {}).get('ms_per_step')))
sys.stdout.flush()
subprocess.call([sys.executable, os.path.join(ROOT, 'tools', 'aggregates.py')])
