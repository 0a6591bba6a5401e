"""Static side of the scalar-cache rule (tools/smem_audit.py): in the objects the library is
linked from, no kernel reads anything but its argument segment through the scalar unit unless it invalidates the
scalar data cache first.  hipcc cross-compiles without a GPU, so this runs in the CPU suite."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_no_scalar_load_of_mutable_memory_without_invalidate():
    from xinvert_amd import build as xbuild
    xbuild.build()                                       # (no-op when the objects are current)
    import smem_audit
    rows, bad = smem_audit.audit(os.path.join(ROOT, 'build', 'obj'))
    assert len(rows) > 200, 'the audit saw only %d kernels' % len(rows)
    assert not bad, 'scalar loads outside the argument segment without s_dcache_inv first: %r' % bad[:5]
    # the only kernels that use the scalar unit for solver data are the ones written to (k_pipe2d, k_pipe3d,
    # k_fusedbih: per-row records, behind their s_dcache_inv)
    users = sorted({name.split('<')[0].replace('void ', '') for _, name, d, ok in rows if d['nonkarg']})
    assert users == ['k_fusedbih', 'k_pipe2d', 'k_pipe3d'], users
