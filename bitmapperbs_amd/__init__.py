"""bitmapperbs_amd -- MI355X-native (gfx950, HIP) bisulfite read-mapping hot path behind BitMapperBS's
--search interface.  See DESIGN.md.  The compute lives in libbmbs_hip.so (C-ABI: include/bmbs.h)."""
__all__ = ["capi", "mapper", "synth"]
