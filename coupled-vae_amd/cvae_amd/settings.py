"""Process-wide switches of the host mirror (the reference keeps these in config.Config)."""
import os


class _Settings(object):
    use_lstm = False          # config.use_lstm (CV/config.py:36); read when a module is constructed
    n_sample_chunk = 2        # _n_sample (CC/Modules/vae_couple.py:20): eps is drawn in (2,B,Z) chunks
    iw_max_rows = 16384       # rows (samples x sentences) decoded per pass in IW evaluation
    sort_eval = True          # IW eval: sentence-major rows ordered by length, finished rows are skipped (exact)
    pair_encoders = True      # dialogue model: post and response encoders in the same launches (ops.encoder_pair)
    ce_save_logits = True     # training: keep the live rows' logits from the CE forward for its backward (no recompute)
    ce_save_logits_max_bytes = 8 << 30   # per call; above it the backward recomputes
    ce_pair = True            # training: the decoder's and the coupled decoder's losses over the same targets share
    #                           one row scan, one zero-fill and (small batches) their gradient products' launches
    shard_samples = False     # multi-GPU IW eval: shard the K samples of every sentence over ranks (dist_utils)
    truncate_eval = True      # cut T to max(len)+1 in test_lm (exact: pad targets weigh 0)
    sort_train_rows = True    # training forward: process the batch's rows in order of non-increasing length
    #                           (modules.RowOrder; per-row outputs come back in the caller's order)
    restore_row_order = True  # False: per-row outputs of forward() stay in the sorted order (callers that only
    #                           take batch means, i.e. harness.Trainer, save the un-permute launches)
    dp_bucket_bytes = int(os.environ.get('CVAE_DP_BUCKET_MB', '16')) << 20   # (env CVAE_DP_BUCKET_MB)
    # data-parallel training: the gradient arena is all-reduced in buckets of about this size,
    #                              each issued (async, on the collective's own stream) as soon as the backward pass has
    #                              produced every gradient in it; 0: one blocking all-reduce after the backward pass
    dp_persist_headroom = int(os.environ.get('CVAE_DP_HEADROOM_CUS', '0'))
    # data-parallel training: CUs the whole-sequence recurrent kernels leave free for RCCL's kernels
    #                              (cvae_persist_headroom).  Off by default: with any head-room the decoder pair (256
    #                              blocks) runs as two launches of 128 and the step takes 6.43 instead of 5.31 ms
    #                              (measured, one GPU); a whole-sequence block fills its CU's register file, so a
    #                              collective's kernel never shares a CU with one either way -- it runs before or
    #                              after, the recurrent kernel's blocks wait for theirs (bounded spins, ~0.25 s), and a
    #                              genuine time-out is handled by Trainer.train_step's fallback
    dp_fallback_steps = 200   # data-parallel training: steps on the per-timestep kernels after a time-out, before the
    #                           whole-sequence kernels are tried again (harness.Trainer.train_step)


settings = _Settings()
