cd $GRAFT_REPO_ROOT
B=tools/ubench/build
echo "lb6 rows376:"; python tools/raster_phases.py 2>&1 | tail -1
echo "lb6 rows250:"; MOOG_RASTER_ROWS=250 python tools/raster_phases.py 2>&1 | tail -1
echo "lb7 rows250:"; MOOG_HIP_LIB=$PWD/$B/libmoog_lb7.so MOOG_RASTER_ROWS=250 python tools/raster_phases.py 2>&1 | tail -1
echo "lb8 rows250:"; MOOG_HIP_LIB=$PWD/$B/libmoog_lb8.so MOOG_RASTER_ROWS=250 python tools/raster_phases.py 2>&1 | tail -1
echo "lb8 rows184:"; MOOG_HIP_LIB=$PWD/$B/libmoog_lb8.so MOOG_RASTER_ROWS=184 python tools/raster_phases.py 2>&1 | tail -1
echo "lb8 rows200:"; MOOG_HIP_LIB=$PWD/$B/libmoog_lb8.so MOOG_RASTER_ROWS=200 python tools/raster_phases.py 2>&1 | tail -1
