"""seedvc_amd -- MI355X-native (gfx950) hot path for seed-vc voice conversion inference.

The directory is named `seed-vc_amd/` (not importable as-is); `__graft_entry__.load_package()`
registers it as the module `seedvc_amd`.
"""
__all__ = ["specs", "weights"]
