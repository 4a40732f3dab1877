for w in 0 16 15 18; do echo "window $w"; python bench.py --emulate-world 8 --steps 12 --warmup 4 $( [ $w != 0 ] && echo --window-bits $w ) 2>/dev/null | tail -1; done
for w in 0 16; do echo "world 4 window $w"; python bench.py --emulate-world 4 --steps 12 --warmup 4 $( [ $w != 0 ] && echo --window-bits $w ) 2>/dev/null | tail -1; done
