"""hipGraph replay of the training step (build extension; the reference launches eagerly).

One `dis_update` + `gen_update` is ~5 000 kernel launches on four streams, enqueued by ~1 500 autograd Functions: 32-41 ms
of host time per step, per rank.  `GraphedStep` runs the step eagerly a few times (every cache -- layer plans, prepared
weight images, workspaces, occupancy queries -- is warm afterwards), captures one more pass into a hipGraph through
`torch.cuda.graph` (the branch streams and the backward-weight side stream fork from and re-join the capture stream, so
the whole multi-stream schedule is part of the graph) and from then on replays it: the host cost of a step becomes a few
small copies and one graph launch.  Device time is HIGHER than the eager three-stream schedule's (see below: the
generator update loses its a / b overlap inside a capture), so this is for host-bound situations, not the default.

What changes between replays lives in device memory: the batch (static input buffers, copied into before the replay),
and the two step-dependent scalars of each Adam update (`FusedAdam.dyn`, refreshed from the host: the learning-rate
schedule and the step counters stay on the host exactly as in the eager path).  Results are bit-identical to eager
steps (tests/test_gpu_graph.py).

Not covered: `optimizer: extra*` (two alternating update kinds), `guided: 0` (fresh host noise per step), a comet logger,
and a process group (the RCCL all-reduce inside a capture has not been exercised on this pool): the constructor raises.
"""
import torch

from . import ops


class GraphedStep:
    def __init__(self, trainer, hyperparameters, x_a, x_b, mask_a=None, mask_b=None, warmup=2):
        import torch.distributed as dist
        hp = hyperparameters
        if "extra" in hp.get("optimizer", "adam"):
            raise NotImplementedError("GraphedStep: ExtraAdam alternates two update kinds; use the eager step")
        if trainer.guided != 1:
            raise NotImplementedError("GraphedStep: guided == 0 draws fresh host noise every step; use the eager step")
        if dist.is_available() and dist.is_initialized():
            raise NotImplementedError("GraphedStep: single-process only (an RCCL all-reduce inside a capture is untested here)")
        if not x_a.is_cuda:
            raise RuntimeError("GraphedStep needs device tensors")
        self.trainer, self.hp = trainer, hp
        self.static = [None if t is None else ops.nhwc(t.detach().clone()) if t.dim() == 4 and t.shape[1] > 1
                       else t.detach().clone() for t in (x_a, x_b, mask_a, mask_b)]
        self._opts = (trainer.dis_opt, trainer.gen_opt)

        from . import trainer as trainer_mod

        def updates():
            trainer.dis_update(self.static[0], self.static[1], hp)
            # gen_update is captured WITHOUT the a / b branch streams (the backward-weight side stream stays): its
            # backward carries gradients across the two branches (c_a, c_b, the style codes), and the cross-stream
            # dependencies the autograd engine then records crash hipStreamEndCapture on ROCm 7.2 (bisected with
            # tools/graph_probe.py: dis_update, whose two discriminators never exchange gradients, captures fine).
            saved = trainer_mod.BRANCH_STREAMS
            trainer_mod.BRANCH_STREAMS = False
            try:
                trainer.gen_update(self.static[0], self.static[1], hp, self.static[2], self.static[3])
            finally:
                trainer_mod.BRANCH_STREAMS = saved

        for _ in range(warmup):                      # eager: fills every cache the capture must not touch
            trainer.update_learning_rate()
            updates()
        torch.cuda.synchronize(x_a.device)
        for opt in self._opts:
            opt.dyn = torch.zeros(2, dtype=torch.float32, device=x_a.device)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):       # nothing executes here: the launches are recorded
                updates()
        except Exception:
            for opt in self._opts:
                opt.dyn = None
            raise

    def __call__(self, x_a, x_b, mask_a=None, mask_b=None):
        """One update_learning_rate + dis_update + gen_update on the given batch; losses land in trainer.loss_*."""
        self.trainer.update_learning_rate()
        for dst, src in zip(self.static, (x_a, x_b, mask_a, mask_b)):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        for opt in self._opts:
            opt.advance_dynamic()
        self.graph.replay()

    def release(self):
        """Back to eager stepping (drops the graph and its memory pool)."""
        for opt in self._opts:
            opt.dyn = None
        self.graph = None
