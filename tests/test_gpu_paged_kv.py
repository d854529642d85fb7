"""GPU: the talker's PAGED KV cache (north star: "paged KV attention at seqlen-1"; include/fq3hip.h fq3_kv_pool_* / fq3_ctx_create_pooled /
fq3_kv_reserve / fq3_kv_release / fq3_kv_adopt).  The reference's analogue is a contiguous StaticCache of max_seq_len slots per graph
object (/root/reference/faster_qwen3_tts/talker_graph.py:43,153-170); here the cache is a pool of 64-key blocks addressed through a
per-context block table that the decode attention kernels, the prefill's KV write and both prefill attention kernels read.

Checked: a pooled context decodes the very ids of a private (static) one although its blocks are scattered over a fragmented pool;
block accounting (idle = 0 blocks, prefilled = the prompt's, armed = prompt + max_new_tokens, released = 0); a staged prompt is
adopted by another context of the pool WITHOUT moving KV rows (block counts move, the ids that follow are those of a direct prefill);
adoption across pools copies; a short pool refuses with FQ3_ENOMEM and takes nothing; the lock-step batch over pooled lanes equals
single-stream runs, and a lane that finished -- whose blocks were returned and re-used by another context -- never writes again."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fq3hip.config import tiny_test_config
from fq3hip.weights import synth_weights, synth_prompt

from test_gpu_batch import _utterance, _arm, _alone            # the single-stream reference helpers of the batch tests


def _mk(cfg, W, dtype, pool=None, share=None, max_seq=200, max_frames=64):
    from fq3hip.engine import Fq3Engine
    return Fq3Engine(cfg, W, device="cuda", dtype=dtype, max_seq_len=max_seq, max_frames=max_frames, share=share, pool=pool)


def _fragment(pool, cfg, W, dtype, first):
    """Leave the pool's free list scrambled: three helper contexts take blocks, the middle one keeps them."""
    a, b, c = (_mk(cfg, W, dtype, pool=pool, share=first) for _ in range(3))
    a.kv_reserve(130); b.kv_reserve(70); c.kv_reserve(190)
    a.kv_release(); c.kv_release()
    return b                                                     # holds 2 blocks from the middle of the id range


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pooled_context_decodes_like_a_private_one(dtype):
    from fq3hip.engine import Fq3KvPool
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, dtype)
    u = _utterance(cfg, dtype, 31, 150, 3, 14, 14, True)         # 150-row prompt: three key tiles, left padding, sampled
    private = _mk(cfg, W, dtype)
    ref, ref_done = _alone(private, cfg, u, 16)
    pool = Fq3KvPool(cfg, 12, dtype=dtype)
    keep = _fragment(pool, cfg, W, dtype, private)
    eng = _mk(cfg, W, dtype, pool=pool, share=private)
    assert eng.kv_blocks() == 0 and pool.stats()["free"] == 12 - 2
    for graph in (False, True):
        _arm(eng, cfg, u)
        assert eng.kv_blocks() == Fq3KvPool.blocks_for(150 + 14 + 1)
        if graph:
            eng.graph_capture()
        else:
            eng.graph_reset()
        eng.decode_frames(16)
        n, done = eng.decode_poll()
        assert n == ref.shape[0] and done == ref_done
        assert torch.equal(eng.decode_codes(0, n).cpu(), ref), f"pooled context (graph={graph}) differs from the private one"
        # the prompt's K / V rows read back through the block table equal the private cache's
        for layer in (0, cfg.talker.num_hidden_layers - 1):
            k1, v1 = eng.kv_export(layer, 150)
            k2, v2 = private.kv_export(layer, 150)
            assert torch.equal(k1[:, 3:], k2[:, 3:]) and torch.equal(v1[:, 3:], v2[:, 3:])
        eng.kv_release()
        assert eng.kv_blocks() == 0
    st = pool.stats()
    assert st["free"] == 12 - 2 and st["high_water"] >= 2 + 3, st
    keep.close(); eng.close(); private.close()


def test_short_pool_refuses_and_takes_nothing():
    from fq3hip.engine import Fq3KvPool
    from fq3hip._lib import Fq3Error, FQ3_ENOMEM
    cfg = tiny_test_config()
    dtype = torch.float32
    W = synth_weights(cfg, 0, dtype)
    first = _mk(cfg, W, dtype)
    pool = Fq3KvPool(cfg, 2, dtype=dtype)
    eng = _mk(cfg, W, dtype, pool=pool, share=first)
    tie, *_ = synth_prompt(cfg, 150, 4, 0, dtype=dtype)
    with pytest.raises(Fq3Error) as ei:
        eng.prefill((tie * 30).to(dtype)[0].cuda().contiguous())          # needs 3 blocks
    assert ei.value.code == FQ3_ENOMEM and "KV pool exhausted" in str(ei.value)
    assert eng.kv_blocks() == 0 and pool.stats()["free"] == 2
    eng.kv_reserve(128)
    assert eng.kv_blocks() == 2
    with pytest.raises(Fq3Error):
        eng.kv_reserve(129)
    assert eng.kv_blocks() == 2
    eng.kv_release(64)
    assert eng.kv_blocks() == 1 and pool.stats()["free"] == 1
    eng.close(); first.close()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adoption_hands_blocks_over_without_copying(dtype):
    """Staged admission: the prompt is prefilled into context A; B adopts it.  Same pool: A's blocks become B's (A owns none
    afterwards, the pool's free count does not move) and B decodes the ids of a direct prefill.  Different pools: rows are copied."""
    from fq3hip.engine import Fq3KvPool
    cfg = tiny_test_config()
    W = synth_weights(cfg, 0, dtype)
    u = _utterance(cfg, dtype, 41, 139, 0, 12, 12, False)
    private = _mk(cfg, W, dtype)
    ref, _ = _alone(private, cfg, u, 12)
    pool = Fq3KvPool(cfg, 10, dtype=dtype)
    other = Fq3KvPool(cfg, 6, dtype=dtype)
    keep = _fragment(pool, cfg, W, dtype, private)
    greedy = dict(temperature=1.0, top_k=0, top_p=1.0, do_sample=False)
    V = cfg.talker.vocab_size
    for dst_pool in (pool, other):
        A = _mk(cfg, W, dtype, pool=pool, share=private)
        B = _mk(cfg, W, dtype, pool=dst_pool, share=private)
        B.kv_reserve(64)                                            # a previous tenant's block: returned by the hand-over
        x = u["tie"][0].cuda().contiguous()
        A.kv_reserve(139 + 12 + 1)                                  # what the scheduler does before it queues the prefill
        logits, hidden = A.prefill(x)
        tok = A.sample(logits, sup_lo=max(0, V - 1024), sup_hi=V, keep_id=cfg.codec_eos_token_id, suppress_eos=True, **greedy)
        free_before, a_blocks = pool.stats()["free"], A.kv_blocks()
        assert a_blocks == Fq3KvPool.blocks_for(152)
        B.kv_adopt(A, 139)
        if dst_pool is pool:
            assert A.kv_blocks() == 0 and B.kv_blocks() == a_blocks
            assert pool.stats()["free"] == free_before + 1          # only B's old block came back; nothing else moved
        else:
            assert A.kv_blocks() == a_blocks and B.kv_blocks() == Fq3KvPool.blocks_for(139)
        B.set_generation_state(0, 0)
        B.set_predictor_sampling(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
        B.decode_begin(first_token=int(tok), prefill_len=139, gen_step=0, past_hidden=hidden, trailing_text=u["tth"][0].cuda().contiguous(),
                       tts_pad_embed=u["tpe"].view(-1).cuda().contiguous(), repetition_penalty=1.0, min_new_tokens=12, max_new_tokens=12, **greedy)
        B.graph_reset()
        B.decode_frames(12)
        n, _ = B.decode_poll()
        assert n == 12 and torch.equal(B.decode_codes(0, n).cpu(), ref), "adopted prompt decodes differently from a direct prefill"
        A.close(); B.close()
    keep.close(); private.close()


def test_pooled_lanes_equal_single_stream_and_finished_lanes_stay_out_of_the_cache():
    """Lock-step batch over lanes of one pool: ids per lane = single-stream ids (VALU GEMVs, fp32).  Lane 1 stops at its budget after
    5 frames; its blocks are released and immediately taken by ANOTHER context, which is prefilled into them while the batch
    keeps decoding: the finished lane must not write into those rows (the reference KV of that context is compared afterwards)."""
    from fq3hip.engine import Fq3KvPool, Fq3Batch
    cfg = tiny_test_config()
    dtype = torch.float32
    W = synth_weights(cfg, 0, dtype)
    utts = [_utterance(cfg, dtype, 51, 70, 0, 14, 14, True), _utterance(cfg, dtype, 52, 40, 2, 5, 5, True),
            _utterance(cfg, dtype, 53, 129, 0, 14, 14, False)]
    private = _mk(cfg, W, dtype)
    ref = [_alone(private, cfg, u, 16) for u in utts]
    pool = Fq3KvPool(cfg, 9, dtype=dtype)
    lanes = [_mk(cfg, W, dtype, pool=pool, share=private) for _ in range(4)]          # the 4th lane is never begun
    batch = Fq3Batch(lanes)
    batch.set_option("mfma", 0)
    for e, u in zip(lanes, utts):
        _arm(e, cfg, u)
    assert [e.kv_blocks() for e in lanes] == [2, 1, 3, 0]
    batch.graph_capture()
    batch.frames(8)
    n1, d1 = lanes[1].decode_poll()                                 # synchronises: lane 1 is over (budget 5)
    assert n1 == 5 and d1
    lanes[1].kv_release()
    spare = _mk(cfg, W, dtype, pool=pool, share=private)
    tie, *_ = synth_prompt(cfg, 60, 4, 0, dtype=dtype, seed=77)
    x = (tie * 30).to(dtype)[0].cuda().contiguous()
    spare.prefill(x)                                                # takes the block lane 1 just returned (LIFO free list)
    k_before, v_before = spare.kv_export(0, 60)
    batch.frames(8)                                                 # lane 1 idles: done lanes leave the cache alone
    for i, (e, (codes, done)) in enumerate(zip(lanes[:3], ref)):
        n, d = e.decode_poll()
        assert n == codes.shape[0] and d == done, (i, n, d)
        assert torch.equal(e.decode_codes(0, n).cpu(), codes), f"lane {i} ids differ from the single-stream run"
    k_after, v_after = spare.kv_export(0, 60)
    assert torch.equal(k_before, k_after) and torch.equal(v_before, v_after), "a finished lane wrote into blocks it no longer owns"
    n, d = lanes[3].decode_poll()
    assert n == 0 and d
    batch.close()
    for e in lanes + [spare, private]:
        e.close()


@pytest.mark.parametrize("graph", [False, True])
def test_a_finished_single_stream_loop_stays_out_of_the_cache(graph):
    """(round-4 advisor) The single-stream fused loop over a POOLED context: once the loop is done (budget reached), frames that were
    queued ahead keep launching its attention kernels; after fq3_kv_release its blocks may belong to another context, so a done loop
    must not append K / V rows (attn_decode_kernel reads DecodeState::done).  Another context is prefilled into the returned block and
    its rows are compared after the idle frames -- direct launches and hipGraph replay."""
    from fq3hip.engine import Fq3KvPool
    cfg = tiny_test_config()
    dtype = torch.float32
    W = synth_weights(cfg, 0, dtype)
    private = _mk(cfg, W, dtype)
    pool = Fq3KvPool(cfg, 4, dtype=dtype)
    A = _mk(cfg, W, dtype, pool=pool, share=private)
    u = _utterance(cfg, dtype, 91, 40, 0, 5, 5, True)
    _arm(A, cfg, u)
    if graph:
        A.graph_capture()
    else:
        A.graph_reset()
    A.decode_frames(8)
    n, d = A.decode_poll()
    assert n == 5 and d
    assert A.kv_blocks() == 1
    A.kv_release()
    spare = _mk(cfg, W, dtype, pool=pool, share=private)
    tie, *_ = synth_prompt(cfg, 60, 4, 0, dtype=dtype, seed=78)
    spare.prefill((tie * 30).to(dtype)[0].cuda().contiguous())      # takes the block A just returned (LIFO free list)
    k_before, v_before = spare.kv_export(0, 60)
    A.decode_frames(8)                                              # frames "queued ahead" of a loop that is over
    torch.cuda.synchronize()
    k_after, v_after = spare.kv_export(0, 60)
    assert torch.equal(k_before, k_after) and torch.equal(v_before, v_after), "a finished loop wrote into blocks it no longer owns"
    for e in (A, spare, private):
        e.close()


def test_scheduler_with_a_short_pool_postpones_requests_and_matches_a_full_pool():
    """BatchDecoder over a pool that cannot hold every request at once: requests wait for blocks instead of failing, every
    utterance completes, with the ids the same scheduler produces over a full-size pool (fp32: lanes are exact)."""
    from types import SimpleNamespace
    from fq3hip.batching import BatchDecoder, BatchRequest
    from fq3hip.engine import Fq3KvPool
    cfg = tiny_test_config()
    dtype = torch.float32
    W = synth_weights(cfg, 0, dtype, parts=("talker", "predictor"))
    first = _mk(cfg, W, dtype, max_frames=32)
    talker = SimpleNamespace(rope_deltas=None)
    tcfg = SimpleNamespace(codec_eos_token_id=cfg.codec_eos_token_id, vocab_size=cfg.talker.vocab_size)
    kw = dict(max_new_tokens=10, min_new_tokens=10, temperature=1.0, top_k=0, top_p=1.0, do_sample=False, repetition_penalty=1.0)
    prompts = [synth_prompt(cfg, 30 + 23 * i, 4, 0, dtype=dtype, seed=200 + i) for i in range(7)]

    def run(blocks):
        pool = Fq3KvPool(cfg, blocks, dtype=dtype)
        mk = lambda: _mk(cfg, W, dtype, pool=pool, share=first, max_frames=32)
        dec = BatchDecoder([mk() for _ in range(3)], predictor_policy=dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0),
                           staging=[mk() for _ in range(3)])
        reqs = [BatchRequest(i, talker, (tie * 30).to(dtype).cuda(), tam.cuda(), tth.cuda(), tpe.cuda(), tcfg, dict(kw))
                for i, (tie, tam, tth, tpe, _r) in enumerate(prompts)]
        out = {}
        for rid, codes, timing in dec.run(reqs):
            assert codes is not None, timing
            out[rid] = codes.cpu()
        st = pool.stats()
        assert st["free"] == st["blocks"], "blocks leaked"
        return out, st

    full, st_full = run(6 * 4)
    short, st_short = run(5)                                          # the largest request needs 3 blocks (168 + 11 slots)
    assert st_short["high_water"] <= 5 < st_full["high_water"]
    assert sorted(full) == sorted(short) == list(range(7))
    for i in range(7):
        assert full[i].shape[0] == 10 and torch.equal(full[i], short[i]), i
