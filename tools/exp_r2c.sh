V=orphics_amd/variants
p() { python tools/r2c_stage_probe.py $1 2>/dev/null | grep stage; }
echo "noload f64 nopf / pf, f32 pf / nopf"; ORPHICS_AMD_LIB=$V/liborphics_amd_noload.so p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_noload.so OA_RS4096_PF=1 p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_noload.so p f32; ORPHICS_AMD_LIB=$V/liborphics_amd_noload.so OA_RS4096_PF=0 p f32
echo "opaque f64 nopf / pf, f32 pf / nopf"; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so OA_RS4096_PF=1 p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so p f32; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so OA_RS4096_PF=0 p f32
echo "default f64, f32"; p f64; p f32
