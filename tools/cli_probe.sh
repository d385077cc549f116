cd vgen_amd
run() { echo "== $*"; timeout -k 5 40 "$@" > /tmp/o.txt 2> /tmp/e.txt; echo "exit $? stdout: $(head -c 200 /tmp/o.txt | tr '\n' ' ') stderr-tail: $(tail -2 /tmp/e.txt | tr '\n' ' ' | cut -c1-200)"; }
run ./vgen-hip range --puzzle 10 -p '^1BgGZ9tcN4rm9KBzDn7KprQz87SZ26SAMH$' -o minimal --gpu-batch-size 8192
run ./vgen-hip generate -p '^1Cat' --no-gpu
run ./vgen-hip range -p boha:b1000:1 -o minimal --gpu-batch-size 8192
run ./vgen-hip estimate -p '^1Cat'
run ./vgen-hip range --range 1:FFF -p '^1O0' -c 0 --gpu-batch-size 8192
run ./vgen-hip range -p boha:b1000:1 -o minimal --gpu-batch-size 8192 --checkpoint /tmp/p.ckpt
