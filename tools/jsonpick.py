import sys, json
d = json.loads(sys.stdin.read())
c = d["config"]
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 3), "ms", round(d["value"], 1), "GF/s levels", c["levels"], "sup", c["supernodes"],
      "nnzL", c["entries_in_factors"], "sweep ms", None if d["roofline"]["seconds_per_launch"] is None else round(d["roofline"]["seconds_per_launch"] * 1e3, 3),
      "first", round(c["first_factorization_s"], 2))
