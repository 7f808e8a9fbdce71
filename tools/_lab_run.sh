run() { echo "### $*"; timeout -k 10 60 "$@" | grep "no stamps\|^conv" ; }
for cfg in "480 854 64 64" "240 427 64 128" "240 427 128 128" "120 214 128 256" "120 214 256 256" "60 107 256 512" "60 107 512 512" "30 54 512 512"; do run build/igemm_lab $cfg 30 || exit 1; done
