# build a -DSHL_DR_TRACE=1 dwpw_resident object, link it with the production objects, swap the library in, trace, swap back
set -e
L=csi-nn2_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -Iinclude -Icsi-nn2_amd/csrc -DSHL_DR_TRACE=1 ${DR_EXTRA} -c csi-nn2_amd/csrc/dwpw_resident.hip -o /tmp/dr_trace.o
objs=$(ls $L/obj/*.o | grep -v dwpw_resident.o)
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/dr_trace.o -o /tmp/libshl_drtrace.so
cp $L/libshl_mi355x.so /tmp/prod.so
cp /tmp/libshl_drtrace.so $L/libshl_mi355x.so
timeout 200 python tools/dev/dr_trace.py --cin 256 --hw 28 --batch 128 2>&1 | tail -20 || true
[ -n "$DR_SKIP512" ] || timeout 200 python tools/dev/dr_trace.py --cin 512 --hw 14 --batch 128 2>&1 | tail -20 || true
cp /tmp/prod.so $L/libshl_mi355x.so
