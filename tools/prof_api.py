import cProfile, pstats, sys, os, io
sys.argv = ["api_case.py", sys.argv[1]] + sys.argv[2:]
pr = cProfile.Profile()
pr.enable()
exec(open("tools/api_case.py").read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue())
