for a in 0 1 2 4 6; do echo "ablate=$a (1: no unit loop, 2: no copy out, 4: no LDS adds)"; PYNAMA_HO3_ABLATE=$a python tools/ho3_case.py ${1:-3} ${2:-64} 3 2>&1 | grep -E "zero blocks"; done
