run() { echo "### TILE=$FOSVOS_FORCE_TILE KS=$FOSVOS_FORCE_KS $*"; timeout -k 10 60 "$@" | grep -v "start times\|lifetime" ; }
export FOSVOS_IGEMM_P=0
for cfg in "480 854 64 64" "240 427 64 128" "240 427 128 128" "120 214 128 256" "120 214 256 256" "60 107 512 512"; do run build/igemm_lab $cfg 30 || exit 1; done
