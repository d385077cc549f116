#!/bin/bash
# Every command line of the reference's README (README.md:50-156), run through vgen-hip on the GPU box: exit code, scan time as the
# tool reports it, wall time of the process, first lines of the output.  usage: bash tools/readme_walk.sh > profiles/rNN_readme_walk.txt
V=vgen_amd/vgen-hip
run() {
    echo "\$ vgen-hip $*"
    local t0=$(date +%s.%N)
    "$V" "$@" > /tmp/walk.out 2> /tmp/walk.err
    local rc=$?
    local t1=$(date +%s.%N)
    echo "  rc=$rc wall=$(python3 -c "print(f'{$t1-$t0:.3f}')") s"
    head -c 700 /tmp/walk.out | sed 's/^/  | /'
    [ -s /tmp/walk.err ] && head -c 400 /tmp/walk.err | sed 's/^/  ! /'
    echo
}
run generate -p "^1Cat"
run generate -p "^1cat" -i
run generate -p 'dead$' -f p2wpkh
run generate -p "^bc1p.*cat" -f p2tr
run generate -p "^3Cat" -f p2sh-p2wpkh
run generate -p "^0xdead" -f ethereum
run generate -p "^1Cat" --no-gpu
run generate -p "^1Cat" --backend vulkan
run generate -p "^1Cat" -c 5 -o minimal
run generate -p "^1Cat" -q
run estimate -p "^1CatDog"
run range --puzzle 66 -p "."
run range -r "20000000000000000:3FFFFFFFFFFFFFFFF"
run generate -p "boha:b1000:66" -l 6
run range -p "boha:b1000:20"
run range -p "boha:b1000:66" -l 8
run verify -k "5HueCGU8rMjxEXxiPuD5BDku4MkFqeZyd4dZ1jvhTVqvbTLvyTJ"
run verify -k "0c28fca386c7a227600b2fe50b7cae11ec86d3bf1fbe471be89827e19d72aa1d"
run verify -k "5HueCGU8rMjxEXxiPuD5BDku4MkFqeZyd4dZ1jvhTVqvbTLvyTJ" -a "1GAehh7TsJAHuUAeKZcXf5CnwuGuGgyX2S"
run list-gpus
run generate -p "^1Cat" -o text
run generate -p "^1Cat" -o json
run generate -p "^1Cat" -o jsonl
run generate -p "^1Cat" -o csv
run generate -p "^1Cat" -o minimal
run generate -p "^1Cat" -o jsonl --file /tmp/results.jsonl
run generate -p "^1Cat" -o csv --file /tmp/results.csv
echo "\$ cat /tmp/results.jsonl /tmp/results.csv"; cat /tmp/results.jsonl /tmp/results.csv | sed 's/^/  | /'
