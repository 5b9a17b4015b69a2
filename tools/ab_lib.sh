for lib in "" nightlight_amd/libnlstack_noslp.so; do
  echo "== lib: ${lib:-default}"
  export NLSTACK_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset NLSTACK_LIB
  tools/qb.sh "--no-cpu" "--no-cpu --mode 3" "--no-cpu --frames 512" "--no-cpu --mode 3 --frames 512 --height 512 --image-height 4096" "--no-cpu --mode 4" "--no-cpu --mode 0" "--no-cpu --frames 300" "--no-cpu --frames 32"
done
