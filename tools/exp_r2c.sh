V=orphics_amd/variants
p() { python tools/r2c_stage_probe.py $1 $2 2>/dev/null | grep stage; }
for rep in 1 2 3; do
echo "default f64 / opaque nopf / opaque pf"; p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so p f64; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so OA_RS4096_PF=1 p f64
done
echo "f32 default / opaque"; p f32; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so p f32
echo "4096 f32 default / opaque; f64 default / opaque"; p f32 4096; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so p f32 4096; p f64 4096; ORPHICS_AMD_LIB=$V/liborphics_amd_opaque.so p f64 4096
