"""Every fast path of the step kernel against its general counterpart.

__graft_entry__.build() also compiles two diagnostic variants of the library in which the fast paths are
disabled or stressed (whole-table scans instead of ring searches and the general two-stage ray instead of the
convex-neighbourhood path, without the outline's miss certificate, with culling boxes rounded by nextafterf and the
determinant's reciprocal as a plain division; the painter walking one sample-grid row per trip instead of four, with the uncertainty
band of its float pre-filter widened 4096-fold so that the float64 confirmation runs constantly, every mask word loaded and
written back instead of the tracked ones, the large parts' observation through the small parts' passes, the stale kd-tree
walked node by node through its queue instead of one lane per node).  The parity suites are run against
each variant in a fresh child process (PAINTRL_LIB selects the library before anything is loaded); the product
build never defines these macros.
"""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('variant', ['force_general_search', 'force_paint_row_trips_wide_band'])
def test_parity_suites_on_forced_general_paths(variant):
    from paintrl_amd import build
    lib = build.variant_path(variant)
    if not os.path.isfile(lib):
        build.build_variant(variant)
    env = dict(os.environ, PAINTRL_LIB=lib)
    suites = [os.path.join(REPO, 'tests', 'test_gpu_parity.py'), os.path.join(REPO, 'tests', 'test_gpu_edge_cases.py')]
    extra = []
    if variant == 'force_paint_row_trips_wide_band':      # its round-5 switches live in the large-part and stale-tree kernels
        suites += [os.path.join(REPO, 'tests', 'test_gpu_big_parts.py'), os.path.join(REPO, 'tests', 'test_gpu_stale_kdtree.py')]
        extra = ['--deselect', 'tests/test_gpu_big_parts.py::test_reference_sized_part_with_the_stale_tree_fits_the_lds',
                 # (the rollout entry points run the same device code as the per-step kernels these suites already drive)
                 '--deselect', 'tests/test_gpu_big_parts.py::test_rollout_entry_points_on_every_configuration',
                 '--deselect', 'tests/test_gpu_big_parts.py::test_fused_rollout_on_a_large_part_with_the_stale_tree',
                 '--deselect', 'tests/test_gpu_stale_kdtree.py::test_stale_tree_in_the_rollout_kernels']
    out = subprocess.run([sys.executable, '-m', 'pytest'] + suites + extra + ['-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-1500:] + out.stderr[-500:]
    assert out.returncode == 0, tail
    assert ' passed' in out.stdout and 'failed' not in out.stdout, tail
