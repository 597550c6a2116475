"""Run by build_ref_binding.sh inside the scratch copy of the patched reference (build container only): the facade hunk of
hip_arch.patch, fast_slic/hip.py, over the reference's own fast_slic/base_slic.py.  Without a GPU the module refuses to import,
exactly as fast_slic/avx2.py:4-8 does without AVX2 -- both outcomes are recorded."""
import cfast_slic

print("archs", cfast_slic.get_supported_archs())
try:
    import fast_slic.hip as h
    print("fast_slic.hip imported:", h.SlicHip.arch_name, h.LSCHip.arch_name, h.on_device(h.SlicHip, 1).arch_name)
except ImportError as e:
    print("fast_slic.hip refused to import:", e)
