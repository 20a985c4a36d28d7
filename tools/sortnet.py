"""Generate and verify (0-1 principle) the per-lane sorting networks of stft_mel.hip (SortNet<R>):
Batcher's odd-even merge sort for the next power of two, pruned to R wires."""


def batcher(n):
    pairs = []

    def merge(lo, n, r):
        m = r * 2
        if m < n:
            merge(lo, n, m)
            merge(lo + r, n, m)
            for i in range(lo + r, lo + n - r, m):
                pairs.append((i, i + r))
        else:
            pairs.append((lo, lo + r))

    def sort(lo, n):
        if n > 1:
            m = n // 2
            sort(lo, m)
            sort(lo + m, m)
            merge(lo, n, 1)

    sort(0, n)
    return pairs


def network(R):
    P = 1
    while P < R:
        P *= 2
    out = []
    for pr in batcher(P):
        if pr[1] < R and (not out or out[-1] != pr):
            out.append(pr)
    return out


def sorts(pairs, R):
    for bits in range(1 << R):
        a = [(bits >> i) & 1 for i in range(R)]
        for i, j in pairs:
            if a[i] > a[j]:
                a[i], a[j] = a[j], a[i]
        if any(a[i] > a[i + 1] for i in range(R - 1)):
            return False
    return True


if __name__ == "__main__":
    for R in (2, 4, 5, 7, 8, 10, 12):
        net = network(R)
        assert sorts(net, R)
        print(R, len(net), net)
