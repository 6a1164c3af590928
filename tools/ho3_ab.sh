for rep in 1 2; do for a in 0 8; do echo "ablate=$a"; PYNAMA_HO3_ABLATE=$a python tools/ho3_case.py ${1:-3} ${2:-64} 3 2>&1 | grep -E "zero blocks"; done; done
