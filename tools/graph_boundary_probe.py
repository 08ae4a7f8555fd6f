"""What does a hipGraph replay cost at its boundaries on this platform?  Two recorded graphs of N small kernels each are replayed
alternately; between them: nothing / an eager kernel / an event record for a side stream / a side-stream kernel behind that record /
a wait on the side stream's event.  Prints the time per pair of replays minus the kernels' own time (one long graph as the yardstick)."""
import sys
import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda")
    x = torch.zeros(1 << 20, device=dev)
    y = torch.zeros(1 << 20, device=dev)
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()

    def record(k):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(k):
                x.add_(1.0)
        return g

    torch.cuda.synchronize()
    g1, g2, g12 = record(n), record(n), record(2 * n)

    def timed(body, reps=200):
        for _ in range(10):
            body()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        e0.record()
        t0 = time.perf_counter()
        for _ in range(reps):
            body()
        host = (time.perf_counter() - t0) / reps * 1e6
        e1.record()
        torch.cuda.synchronize()
        timed.host = host          # the loop's own (enqueue) time per repetition: is the figure the device's or the host's?
        return e0.elapsed_time(e1) / reps * 1e3

    base = timed(lambda: g12.replay())
    print(f"one graph of {2 * n} kernels: {base:7.1f} us per replay   host enqueue {timed.host:6.1f} us")

    def case(name, between):
        def body():
            g1.replay()
            g2.replay()
            between()
        t = timed(body)
        print(f"{name:58s} {t:7.1f} us per pair  (+{t - base:6.1f})   host enqueue {timed.host:6.1f} us")

    main_s = torch.cuda.current_stream()
    case("two graphs back to back", lambda: None)
    case("... + an eager kernel between the pairs", lambda: y.add_(1.0))

    def rec():
        side.wait_stream(main_s)
    case("... + an event record on the main stream (side.wait_stream)", rec)

    def rec_kernel():
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            y.add_(1.0)
    case("... + record, and a kernel on the side stream behind it", rec_kernel)

    state = {}

    def rec_kernel_wait():
        if "ev" in state:
            main_s.wait_event(state["ev"])
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            y.add_(1.0)
            ev = torch.cuda.Event()
            ev.record(side)
        state["ev"] = ev
    case("... + the same, and the main stream waits for the previous one", rec_kernel_wait)

    def eager_then_side():
        y.add_(1.0)
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            for _ in range(5):
                y.add_(1.0)
    case("... + eager kernel, record, five side-stream kernels", eager_then_side)

    # which call is it?  a wait alone; a record that follows an eager kernel instead of a graph; an event recorded by a node of the graph
    ev_side = torch.cuda.Event()
    with torch.cuda.stream(side):
        y.add_(1.0)
        ev_side.record(side)
    case("... + only main.wait_event(an old side-stream event)", lambda: main_s.wait_event(ev_side))

    def eager_record():
        y.add_(1.0)
        side.wait_stream(main_s)
    case("... + eager kernel, then the record", eager_record)

    def record_eager():
        side.wait_stream(main_s)
        y.add_(1.0)
    case("... + the record, then an eager kernel", record_eager)

    def eager_record_eager():
        y.add_(1.0)
        side.wait_stream(main_s)
        y.add_(1.0)
    case("... + eager kernel, record, eager kernel", eager_record_eager)

    try:
        ext = torch.cuda.Event(external=True)
        g2e = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2e, stream=cap):
            for _ in range(n):
                x.add_(1.0)
            ext.record()

        def body_ext():
            g1.replay()
            g2e.replay()
            side.wait_event(ext)
            with torch.cuda.stream(side):
                y.add_(1.0)
        t = timed(body_ext)
        print(f"{'event recorded by a node of the second graph + side kernel':58s} {t:7.1f} us per pair  (+{t - base:6.1f})")
    except Exception as exc:
        print("external event node:", type(exc).__name__, exc)

    # does the cost of a record stay with the stream?  one record every 4th / 16th pair; a record on ANOTHER stream that waited for main
    k = {"i": 0}

    def every(nth):
        def f():
            k["i"] += 1
            if k["i"] % nth == 0:
                side.wait_stream(main_s)
        return f
    case("a record on main every 4th pair", every(4))
    case("a record on main every 16th pair", every(16))
    case("two records on main per pair", lambda: (side.wait_stream(main_s), side.wait_stream(main_s)))
    # (stream memory operations -- hipStreamWriteValue32 / hipStreamWaitValue32 -- would be the event-free hand-over, but
    # hipDeviceAttributeCanUseStreamWaitValue is 0 on this device and hipExtMallocWithFlags(hipMallocSignalMemory) fails)


if __name__ == "__main__":
    main()
