"""
What was a device address to the node store when the process died?

    RUBIKS_VMM_LOG=/path/vmm.log python bench.py ...            # every reserve / map / release, flushed line by line
    python tools/vmm_classify.py /path/vmm.log 0x7e224488d000   # e.g. the address of "Memory access fault by GPU node-2 ... on address"

Every process writes its own file, /path/vmm.log[.rank<R>].<pid> (ranks of one job and child processes inherit the variable; their
address spaces have nothing to do with each other).  Give the file itself, or the prefix when only one process wrote, or the prefix
and --pid <pid>.

Replays the event file of rl-rubiks_amd/csrc/rubiks_vmm.hip (the same bookkeeping as rc_vmm_classify inside a live process) and
prints `<address> <kind> ...`:
    mapped    inside a live range, memory behind the chunk           -> the fault is not the node store's mapping
    unmapped  inside a live range, NO memory behind the chunk        -> a row was touched before it was mapped (a missing guard / growth step)
    slack     inside a live reservation, outside the range handed out (alignment slack, rest of the size class)
    idle      inside a released reservation (nothing mapped)         -> use after release
    unknown   never inside a reservation of this library             -> a wild address (e.g. computed from wrong data) or somebody else's
plus, for idle / unknown, whether the address EVER lay in a range of ours and what happened to that range last.
"""
import sys

CHUNK_MIN = 2 << 20


def replay(path):
    live, idle, pending, history = {}, {}, None, []     # live: base -> dict; idle: raw -> (raw_bytes, uses)
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        seq, op, base, a, b, rc = line.split()
        seq, base, a, b, rc = int(seq), int(base, 16), int(a), int(b), int(rc)
        if op == "A":
            pending = (base, a)
            idle.pop(base, None)
        elif op in "RU":
            raw, raw_bytes = pending
            live[base] = {"raw": raw, "raw_bytes": raw_bytes, "bytes": a, "chunk": b, "uses": rc, "mapped": set(), "seq": seq}
            history.append((seq, op, raw, raw_bytes))
        elif op == "M":
            r = live[base]
            r["mapped"].update(range(a // r["chunk"], (a + b) // r["chunk"]))
        elif op == "X":
            r = live.pop(base)
            history.append((seq, op, r["raw"], r["raw_bytes"]))
        elif op in "IQ":        # Q: a range whose unmap / flush failed -- kept out of circulation for good, nothing is ever mapped there again
            idle[base] = (a, b)
        elif op == "F":
            history.append((seq, op, base, a))
    return live, idle, history


def classify(addr, live, idle, history):
    for base, r in live.items():
        if r["raw"] <= addr < r["raw"] + r["raw_bytes"]:
            if not (base <= addr < base + r["bytes"]):
                return "slack", f"reservation 0x{r['raw']:x}+{r['raw_bytes']} of live range 0x{base:x} (use {r['uses']})"
            c = (addr - base) // r["chunk"]
            kind = "mapped" if c in r["mapped"] else "unmapped"
            return kind, f"live range 0x{base:x}+{r['bytes']} (use {r['uses']}, reserved at event {r['seq']}), chunk {c} of {r['chunk']} bytes, offset {addr - base}"
    for raw, (raw_bytes, uses) in idle.items():
        if raw <= addr < raw + raw_bytes:
            return "idle", f"idle reservation 0x{raw:x}+{raw_bytes} after {uses} use(s)"
    past = [(seq, op) for seq, op, raw, raw_bytes in history if raw <= addr < raw + raw_bytes]
    return "unknown", ("never inside a reservation of the node store" if not past else
                       f"inside a reservation that no longer exists; its last event: {past[-1][1]} at {past[-1][0]}")


if __name__ == "__main__":
    import glob
    import os
    args = sys.argv[1:]
    pid = None
    if "--pid" in args:
        i = args.index("--pid")
        pid = args[i + 1]
        del args[i:i + 2]
    if len(args) < 2:
        sys.exit(__doc__)
    path = args[0]
    if not os.path.isfile(path):
        found = sorted(f for f in glob.glob(path + ".*") if pid is None or f.endswith("." + pid))
        if len(found) != 1:
            sys.exit(f"{path}: {len(found)} event files match ({', '.join(found) or 'none'}); name one, or give --pid")
        path = found[0]
    state = replay(path)
    for text in args[1:]:
        addr = int(text, 16) if text.lower().startswith("0x") else int(text)
        kind, where = classify(addr, *state)
        print(f"0x{addr:x} {kind} -- {where}")
