import torch, sys
a=torch.load(sys.argv[1]); b=torch.load(sys.argv[2])
bad=[k for k in a if not torch.equal(a[k],b[k])]
print(sys.argv[1], sys.argv[2], "differ:", len(bad), "of", len(a), bad[:4], float(a["loss"]), float(b["loss"]))
