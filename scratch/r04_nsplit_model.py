"""Round 4: the scorer's split count - r03's rule (double until 512 workgroups) against the launch model of csrc/score.hip's
choose_nsplit (cost = rounds / ns * (1 + 0.005 ns)).  Prints every shape of the grid on which they differ (none for powers of two)
and the choices for some ragged batch sizes."""
def old(n_r, n_c, target=512):
    nrb = (n_r + 127) // 128
    ns = 1
    while nrb * ns < target and ns * 2 * 64 <= n_c and ns < 64:
        ns *= 2
    return ns
def new(n_r, n_c, rows=128, slots=512):
    nrb = (n_r + rows - 1) // rows
    best, bc = 1, None
    for ns in range(1, 65):
        if ns > 1 and ns * 64 > n_c:
            break
        rounds = (nrb * ns + slots - 1) // slots
        c = rounds / ns * (1 + 0.005 * ns)
        if bc is None or c < bc - 1e-12:
            best, bc = ns, c
    return best
sizes = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144]
diff = [(n, m) for n in sizes for m in sizes if old(n, m) != new(n, m) or old(n, m, 512) != new(n, m, 256, 256)]
print("power-of-two shapes on which the rules differ:", diff)
for n in [8200, 8000, 6000, 4100, 10000, 12288, 16000, 1000, 3000]:
    print(n, "pass 1:", old(n, n), "->", new(n, n), " dc pass:", old(n, n, 512), "->", new(n, n, 256, 256))
