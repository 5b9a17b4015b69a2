for rep in 1 2; do
for n in 128 100 120 60; do
  for lib in libnlstack.so libnlstack_head.so; do
    echo -n "$lib "; NLSTACK_LIB=$PWD/nightlight_amd/$lib python tools/ab_flags.py 2 $n 4096 0 4096 3 0 | cut -c1-100
  done
done
done
for lib in libnlstack.so libnlstack_head.so; do
    echo -n "$lib "; NLSTACK_LIB=$PWD/nightlight_amd/$lib python tools/ab_flags.py 3 100 4096 0 4096 2 0 | cut -c1-100
    echo -n "$lib "; NLSTACK_LIB=$PWD/nightlight_amd/$lib python tools/ab_flags.py 0 100 4096 0 4096 2 0 | cut -c1-100
done
