"""Where the 64-row attention core's cycles go: the product build with parts taken out (lab library, variants 100 + 512 + bits), read
through the in-kernel s_memtime stamps in CYCLES per tile - not microseconds: an ablated build draws different power and the chip
answers with a different clock (round 6 chased a "200-cycle branch" for an hour that was a build computing NaN at a higher clock;
later the same chip ran a perfectly balanced schedule at 1.81 GHz instead of 1.95 and lost what the balance had won).
Bits: 1 no softmax VALU (P = 1), 2 no LDS-DMA, 8 no tile barrier (results of such builds are wrong by construction).
    python tools/attn64_ablate.py"""
import os, re, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
NAMES = {612: "as shipped", 613: "- softmax VALU", 615: "- softmax VALU - DMA", 620: "- tile barrier", 621: "- softmax VALU - barrier", 623: "- softmax VALU - DMA - barrier (MFMAs, fragment reads, loop control)"}
for var, name in NAMES.items():
    r = subprocess.run([sys.executable, os.path.join(here, "attn64_stamps.py"), str(var)], capture_output=True, text=True)
    line = next((l for l in r.stdout.split("\n") if l.startswith("B4 L4096 rects 4")), r.stderr[-300:])
    m = re.search(r"(\d+) cycles per tile", line)
    cyc = m.group(1) if m else "?"
    m = re.search(r"us at ([0-9.]+) GHz", line)
    clk = m.group(1) if m else "?"
    m = re.search(r"tile barrier (\d+)", line)
    bar = m.group(1) if m else "?"
    print(f"{name:75s} {cyc:>6s} cycles per tile (56 MFMAs = 1664 pipe cycles), of them {bar} at the tile barrier (wave 0); clock held {clk} GHz", flush=True)
print("halves of the blind iteration:")
subprocess.run([sys.executable, os.path.join(here, "attn64_halves.py")])
