"""One-process-per-GPU sharded encode/decode over torch.distributed (backend "nccl" = RCCL over xGMI).

SURVEY 8(e): rank g owns input bytes [g*N/G, (g+1)*N/G).  A single global code is needed for a stream
that is bit-exact with the single-stream reference, which costs exactly two tiny exchanges:
  1. all-reduce(sum) of the 256 byte counts (2 KiB)       -> identical tables on every rank
  2. all-gather of the per-rank body bit totals (8 B/rank) -> every rank's global bit offset
Both are latency-bound; everything else is rank-local.  No host synchronisation: the offsets stay in
device memory and the emit kernel reads them there.

`be` is the backend that runs the stages: golden_huffman_amd.ghf.Context in the product; the CPU gloo
tests plug in a stand-in built on the oracle to exercise this file's offset/merge logic without a GPU.
"""

_MAXLEN_OFF = (257 * 3 + 128) * 4 + 4  # byte offset of ghf_code.max_len


def header_bits_of(be, d_code):
    """8 * (1040 + 8 * max_len) as a device int64[1], read from the device-resident tables."""
    torch = be.torch
    max_len = d_code[_MAXLEN_OFF : _MAXLEN_OFF + 4].view(torch.int32).to(torch.int64)
    return (max_len * 8 + 1040) * 8


def encode_sharded(be, dist, d_in, out=None, index=None, group=None):
    """Encode this rank's shard.  Returns a dict:
         out        local output buffer.  rank 0: stream byte 0 onwards (header included);
                    rank g>0: stream byte 16*(start_bit/128) onwards (GHF_EMIT_REBASE)
         start_bit  device int64[1]: absolute stream bit of this shard's first code
         end        device int64[2]: {absolute end bit, defined bytes in `out`}
         code       device tables (identical on all ranks)
         totals     device int64[world]: body bits per rank
    """
    torch = be.torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = d_in.numel()
    hist = be.histogram(d_in)  # K1, local
    if world > 1:
        counts = hist[:256]  # the end-mark slot [256] == 1 must not be summed G times
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    d_code = be.build_code(hist)  # K2+K3 redundantly on every rank: deterministic -> no broadcast
    total = be.encode_plan(d_in, d_code)  # K4, local
    if world > 1:
        totals = torch.empty(world, dtype=torch.int64, device=total.device)
        dist.all_gather_into_tensor(totals, total, group=group)
    else:
        totals = total.clone()
    if hasattr(be, "shard_start_bit"):
        start_bit = be.shard_start_bit(d_code, totals, world, rank)  # one tiny kernel
    else:
        before = totals[:rank].sum().reshape(1) if rank > 0 else torch.zeros(1, dtype=torch.int64, device=total.device)
        start_bit = header_bits_of(be, d_code) + before
    flags = 0
    if rank == world - 1:
        flags |= be.EMIT_LAST
    if rank > 0:
        flags |= be.EMIT_REBASE
    if out is None:
        # a shard is packed with the GLOBAL code, which can be far from optimal for it (a uniform shard among
        # low-entropy ones): compress_bound's 9 bits per symbol only hold for a buffer's own code
        bound = be.shard_bound(n) if (world > 1 and hasattr(be, "shard_bound")) else be.compress_bound(n)
        out = be.empty_u8(bound)
    if index is not None:
        index.flags = 0 if rank == world - 1 else 1  # GHF_INDEX_NO_END_MARK: this shard is not followed by the end mark
    if rank == 0:
        if hasattr(be, "EMIT_HEADER"):
            flags |= be.EMIT_HEADER  # a5: the header rides along with the emit launches
        else:
            be.write_header(d_code, out)
    end = be.encode_emit(d_in, d_code, out, start_bit=start_bit, flags=flags, index=index)  # K5
    return {"out": out, "start_bit": start_bit, "end": end, "code": d_code, "totals": totals, "n": n}


def decode_sharded(be, enc, index, d_out=None):
    """Decode this rank's shard from its own local buffer + side-car (embarrassingly parallel)."""
    return be.decode(enc["out"], int(enc["out"].numel()), enc["code"], index, d_out=d_out)


def gather_stream(be, dist, enc, group=None):
    """Host-side assembly of the complete .crs2 image on every rank (for files / verification only).
    Adjacent ranks share at most one 16-byte unit; both sides wrote only their own bits into zeroed
    memory, so the shared bytes are OR-ed."""
    import numpy as np

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    start_bit = int(enc["start_bit"].item())
    end_bit, nbytes = (int(x) for x in enc["end"].tolist())
    origin = 0 if rank == 0 else (start_bit >> 7) << 4
    local = enc["out"][:nbytes].cpu().numpy()
    pieces = [None] * world
    if world > 1:
        dist.all_gather_object(pieces, (origin, end_bit, local), group=group)
    else:
        pieces[0] = (origin, end_bit, local)
    total_bytes = max((e + 7) // 8 for _, e, _ in pieces)
    stream = np.zeros(total_bytes, dtype=np.uint8)
    for origin, _, buf in pieces:
        stream[origin : origin + buf.size] |= buf
    return stream


def decode_foreign_sharded(be, dist, host_stream, group=None):
    """Decode a .crs2 that has NO side-car (e.g. one the reference wrote) on all ranks ("per-rank
    self-sync + one all-gather of symbol counts").  host_stream: the whole file as a numpy uint8 array, the same on
    every rank (a real deployment would read each rank's byte range only).  The body is cut into world byte ranges;
    rank g re-synchronises on its piece from a guessed first code boundary, tells rank g+1 where its last code ends,
    and the guesses are iterated until none moves (Huffman codes self-synchronise: one or two rounds).  Returns
    (d_out, n_local, out_offset): this rank's decoded bytes, how many, and where they belong in the whole output."""
    import numpy as np

    torch = be.torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    code, hs = be.parse_header(host_stream)
    d_code = be.code_to_device(code)
    body = host_stream[hs:]
    lo, hi = rank * body.size // world, (rank + 1) * body.size // world
    own = hi - lo
    piece = np.zeros(((own + 16 + 15) // 16) * 16 + 16, dtype=np.uint8)  # own bytes + look-ahead, zero behind the stream
    avail = min(body.size, hi + 16) - lo
    piece[:avail] = body[lo : lo + avail]
    d_piece = torch.from_numpy(piece).to(be.device)
    first = 0
    res = None
    for _ in range(world + 1):
        res = be.sync_piece(d_piece, piece.size, first, own * 8, d_code) if own else (0, 0, False)
        landings = [None] * world
        dist.all_gather_object(landings, int(res[0]), group=group)
        new_first = 0 if rank == 0 else landings[rank - 1]
        moved = [None] * world
        dist.all_gather_object(moved, int(new_first != first), group=group)
        first = new_first
        if not any(moved):
            break
    landing, n_local, has_eof = res
    counts = [None] * world
    dist.all_gather_object(counts, (int(n_local), bool(has_eof)), group=group)
    # nothing behind the end mark counts (pieces behind it hold padding / look-ahead only)
    seen_eof = False
    offset = 0
    for g, (cnt, eof) in enumerate(counts):
        if g == rank:
            if seen_eof:
                n_local = 0
            break
        if not seen_eof:
            offset += cnt
        seen_eof = seen_eof or eof
    if n_local == 0:
        return be.empty_u8(1), 0, offset
    d_out, _ = be.decode(d_piece, piece.size, d_code, None, cap=n_local + 64)  # reuses the side-car sync_piece rebuilt
    return d_out, n_local, offset
