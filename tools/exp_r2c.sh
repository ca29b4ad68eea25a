V=orphics_amd/variants
p() { python tools/r2c_stage_probe.py $1 2>/dev/null | grep stage; }
echo "wsync with lgkmcnt(0): f64 f32"; p f64; p f32
echo "compiler fence only: f64 f32"; ORPHICS_AMD_LIB=$V/liborphics_amd_nowait.so p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_nowait.so p f32
echo "again with wait"; p f64; p f32
