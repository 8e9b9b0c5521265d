"""The hot path as the reference drivers run it: CFM sampler -> strip prompt -> vocoder, plus the
chunk / crossfade loop around it (reference: inference.py:470-527, seed_vc_wrapper.py:561-623) and the
multi-GPU sharding of utterance batches.
"""
import numpy as np
import torch


def crossfade(chunk1, chunk2, overlap):
    """cos^2 crossfade on numpy chunks, in place on chunk2 (reference: inference.py:343-350)."""
    fade_out = np.cos(np.linspace(0, np.pi / 2, overlap)) ** 2
    fade_in = np.cos(np.linspace(np.pi / 2, 0, overlap)) ** 2
    if len(chunk2) < overlap:
        chunk2[:overlap] = chunk2[:overlap] * fade_in[:len(chunk2)] + (chunk1[-overlap:] * fade_out)[:len(chunk2)]
    else:
        chunk2[:overlap] = chunk2[:overlap] * fade_in + chunk1[-overlap:] * fade_out
    return chunk2


def stream_wave_chunks(vc_wave, processed_frames, n_target_frames, overlap_wave_len, overlap_frame_len, chunks,
                       previous_chunk, is_last_chunk):
    """One turn of the drivers' chunk state machine (reference: `SeedVCWrapper._stream_wave_chunks`,
    seed_vc_wrapper.py:201-285, non-streaming leg; the same statements inline at inference.py:507-527): appends this
    chunk's share of the output to `chunks` and returns (processed_frames, previous_chunk, should_break).
    vc_wave: (1, L) tensor.  The first chunk gives everything but its last `overlap_wave_len` samples, a middle chunk is
    crossfaded over its first `overlap_wave_len` samples with the tail kept from its predecessor and gives everything
    but its own tail, the last chunk is crossfaded and given whole; `processed_frames` advances by the chunk's frames
    minus `overlap_frame_len` (not at all when the first chunk is also the last, as in the reference)."""
    if processed_frames == 0:
        if is_last_chunk:
            chunks.append(vc_wave[0].cpu().numpy())
            return processed_frames, previous_chunk, True
        chunks.append(vc_wave[0, :-overlap_wave_len].cpu().numpy())
        return processed_frames + n_target_frames - overlap_frame_len, vc_wave[0, -overlap_wave_len:], False
    if is_last_chunk:
        chunks.append(crossfade(previous_chunk.cpu().numpy(), vc_wave[0].cpu().numpy(), overlap_wave_len))
        return processed_frames + n_target_frames - overlap_frame_len, previous_chunk, True
    chunks.append(crossfade(previous_chunk.cpu().numpy(), vc_wave[0, :-overlap_wave_len].cpu().numpy(), overlap_wave_len))
    return processed_frames + n_target_frames - overlap_frame_len, vc_wave[0, -overlap_wave_len:], False


def chunk_plan(n_src, max_source_window, overlap_frame_len):
    """Chunk boundaries of the reference driver's loop (inference.py:473-527): [(first source frame, frames, is_last)].
    They depend on lengths only: each chunk takes up to `max_source_window` frames and the next one starts
    `overlap_frame_len` frames before its end."""
    plan, processed = [], 0
    while processed < n_src:
        s_len = min(max_source_window, n_src - processed)
        is_last = processed + max_source_window >= n_src
        plan.append((processed, s_len, is_last))
        if is_last:
            break
        processed += s_len - overlap_frame_len
    return plan


class HotPath:
    """cfm: seedvc_amd.cfm.CFM ; vocoder: seedvc_amd.vocoder.BigVGAN | HiFT."""

    def __init__(self, cfm, vocoder):
        self.cfm = cfm
        self.vocoder = vocoder

    @torch.inference_mode()
    def convert_batch(self, mu, prompt, style, n_timesteps, inference_cfg_rate, z=None, x_lens=None,
                      prompt_lens=None, vocoder_kwargs=None):
        """B utterances (each an independent reference run): -> (mel (B,C,S), wave (B, S*hop))."""
        B, T = mu.size(0), mu.size(1)
        P = prompt.size(-1)
        lens = x_lens if x_lens is not None else torch.LongTensor([T] * B)
        mel = self.cfm.inference(mu, lens, prompt, style, None, n_timesteps, inference_cfg_rate=inference_cfg_rate,
                                 z=z, prompt_lens=prompt_lens)
        vc_target = mel[:, :, P:]                                   # inference.py:505
        wave = self.vocoder(vc_target.float(), **(vocoder_kwargs or {}))
        return vc_target, wave.reshape(B, -1)

    @torch.inference_mode()
    def convert_long(self, cond, prompt_condition, mel2, style2, n_timesteps, inference_cfg_rate, hop,
                     max_context_window, overlap_frame_len=16, noise_fn=None, vocoder_kwargs_fn=None):
        """One long utterance, chunked exactly like the reference driver (inference.py:470-527): chunks of
        max_context_window - P source frames, advancing by S_chunk - 16, 16-frame cos^2 crossfade on the host.
        noise_fn(T) -> z (1,C,T) lets a caller pin the noise; vocoder_kwargs_fn(S) pins vocoder draws."""
        overlap_wave_len = overlap_frame_len * hop
        P = mel2.size(2)
        max_source_window = max_context_window - P
        processed = 0
        chunks, previous = [], None
        while processed < cond.size(1):
            chunk_cond = cond[:, processed:processed + max_source_window]
            is_last = processed + max_source_window >= cond.size(1)
            cat_condition = torch.cat([prompt_condition, chunk_cond], dim=1)
            T = cat_condition.size(1)
            z = noise_fn(T) if noise_fn is not None else None
            vc_target = self.cfm.inference(cat_condition, torch.LongTensor([T]), mel2, style2, None, n_timesteps,
                                           inference_cfg_rate=inference_cfg_rate, z=z)[:, :, P:]
            kw = vocoder_kwargs_fn(vc_target.size(2)) if vocoder_kwargs_fn is not None else {}
            vc_wave = self.vocoder(vc_target.float(), **kw).reshape(1, -1)
            processed, previous, should_break = stream_wave_chunks(vc_wave, processed, vc_target.size(2), overlap_wave_len,
                                                                   overlap_frame_len, chunks, previous, is_last)
            if should_break:
                break
        return torch.tensor(np.concatenate(chunks))[None, :].float()


    @torch.inference_mode()
    def convert_long_device(self, cond, prompt_condition, mel2, style2, n_timesteps, inference_cfg_rate, hop,
                            max_context_window, overlap_frame_len=16, noise_fn=None, vocoder_kwargs_fn=None):
        """`convert_long` with everything kept on the device: same chunk boundaries, the cos^2
        crossfade done by `svc_crossfade` in the reference's float64 arithmetic (bit-identical), output assembled in one
        device buffer, and the vocoder of chunk k running on a second HIP stream beside the sampler of chunk k+1.
        One host synchronisation at the end instead of two `.cpu()` round trips per chunk."""
        import ctypes as C
        from . import _lib
        dev = cond.device
        ovw = overlap_frame_len * hop
        P = mel2.size(2)
        msw = max_context_window - P
        n_src = cond.size(1)
        plan = chunk_plan(n_src, msw, overlap_frame_len)
        sizes = []
        for k, (_, s_len, is_last) in enumerate(plan):
            full = s_len * hop
            sizes.append(full if is_last else full - ovw)
        out = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        fade_out = torch.from_numpy(np.cos(np.linspace(0, np.pi / 2, ovw)) ** 2).to(dev)
        fade_in = torch.from_numpy(np.cos(np.linspace(np.pi / 2, 0, ovw)) ** 2).to(dev)
        s_main = torch.cuda.current_stream(dev)
        s_voc = Lanes._lane_stream(torch.device(dev), "vocoder")      # one per device for the process (see Lanes)
        s_voc.wait_stream(s_main)
        off, prev_tail, keep = 0, None, []
        for k, (p0, s_len, is_last) in enumerate(plan):
            cat_condition = torch.cat([prompt_condition, cond[:, p0:p0 + s_len]], dim=1)
            T = cat_condition.size(1)
            z = noise_fn(T) if noise_fn is not None else None
            vc_target = self.cfm.inference(cat_condition, torch.LongTensor([T]), mel2, style2, None, n_timesteps,
                                           inference_cfg_rate=inference_cfg_rate, z=z)[:, :, P:]
            kw = vocoder_kwargs_fn(vc_target.size(2)) if vocoder_kwargs_fn is not None else {}
            ready = torch.cuda.Event()
            ready.record(s_main)
            s_voc.wait_event(ready)
            with torch.cuda.stream(s_voc):
                wave = self.vocoder(vc_target.float(), **kw).reshape(-1)
                body = wave if is_last else wave[:-ovw]
                if k > 0:
                    n = min(body.numel(), ovw)
                    _lib.check(_lib.lib().svc_crossfade(_lib.ptr(body), _lib.ptr(prev_tail), _lib.ptr(fade_in),
                                                        _lib.ptr(fade_out), n, _lib.stream_ptr()))
                out[off:off + body.numel()].copy_(body)
                off += body.numel()
                prev_tail = None if is_last else wave[-ovw:]
                keep.append((vc_target, wave))          # alive until the side stream has consumed them
        s_main.wait_stream(s_voc)
        torch.cuda.current_stream(dev).synchronize()
        del keep
        return out[None, :]


# ----------------------------------------------------------------------------------------- multi-GPU sharding
def shard_range(n_items, rank, world_size):
    """Contiguous block partition of `n_items` utterances over ranks (first ranks take the remainder)."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_audio(local_wave, local_lens, n_items, group=None):
    """Gather per-rank output audio on rank 0.  local_wave (n_local, Lmax_local) float32, local_lens list[int].
    One length all-gather (ints) + one padded gather of audio; no collective is used inside the sampler or
    vocoder (utterances are independent).  Returns a list of 1-D tensors on rank 0, None elsewhere."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local_wave[i, :local_lens[i]] for i in range(local_wave.size(0))]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_wave.device
    # lengths: every rank contributes a fixed-size vector (max shard size), -1 padded
    max_shard = (n_items + world - 1) // world
    lens = torch.full((max_shard,), -1, dtype=torch.int64, device=dev)
    lens[:len(local_lens)] = torch.tensor(local_lens, dtype=torch.int64, device=dev)
    all_lens = [torch.empty_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    all_lens = torch.stack(all_lens).cpu()          # one device -> host copy; the loops below index host memory
    Lmax = int(all_lens.max())
    buf = torch.zeros(max_shard, Lmax, dtype=torch.float32, device=dev)
    if local_wave.numel():
        buf[:local_wave.size(0), :local_wave.size(1)] = local_wave
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        for i in range(max_shard):
            n = int(all_lens[r][i])
            if n >= 0:
                out.append(gathered[r][i, :n])
    return out


class Lanes:
    """Several independent (sampler, vocoder) handle pairs on one GPU, each on its own HIP stream.

    The C ABI is re-entrant per (handle, stream) pair; utterances are independent, so a batch is split across lanes
    and the lanes' kernels overlap on the device (one lane's tail waves / small kernels run beside the other's
    GEMMs).  make_pair() -> (cfm, vocoder) is called once per lane."""

    # Lane streams are created once per device and shared by every Lanes object of the process: the runtime maps HIP
    # streams onto a few hardware queues as they are first used, and the streams of a SECOND Lanes object (a service that
    # rebuilds its models; bench.py's secondary workloads) were seen to land on one queue -- its lanes then run one after
    # the other (small B = 64: 40.6 k -> 38.4 k frames/s for whatever ran second in a process).
    _streams = {}

    @classmethod
    def _lane_stream(cls, device, i):
        key = (device.index if device.index is not None else torch.cuda.current_device(), i)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=device)
        return cls._streams[key]

    def __init__(self, make_pair, n_lanes=2, device="cuda:0"):
        self.device = torch.device(device)
        self.lanes = []
        for i in range(n_lanes):
            cfm, voc = make_pair()
            self.lanes.append((HotPath(cfm, voc), self._lane_stream(self.device, i)))

    @torch.inference_mode()
    def convert_batch(self, mu, prompt, style, n_timesteps, inference_cfg_rate, z=None, vocoder_kwargs=None):
        B = mu.size(0)
        n = len(self.lanes)
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        outs, done = [], []
        for i, (hp, st) in enumerate(self.lanes):
            s, e = shard_range(B, i, n)
            if e == s:
                continue
            st.wait_event(start)
            with torch.cuda.stream(st):
                kw = {k: v[s:e] for k, v in (vocoder_kwargs or {}).items()}
                outs.append(hp.convert_batch(mu[s:e], prompt[s:e], style[s:e], n_timesteps, inference_cfg_rate,
                                             z=None if z is None else z[s:e], vocoder_kwargs=kw))
                ev = torch.cuda.Event()
                ev.record(st)
                done.append(ev)
        for ev in done:
            cur.wait_event(ev)
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
