/* sbc_hip.h -- C ABI of libsbc_hip.so: the annealed-Langevin MIMO channel-estimation hot path of
 * utcsilab/score-based-channels as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference has no FFI of its own: its operator boundary is the PyTorch call
 *     score = diffuser(current_real, labels)                 (src/score_based_channels/test_score.py:151)
 * on an NCSNv2Deepest nn.Module (ncsnv2/models/ncsnv2.py:198-300) plus the tensor expressions of the
 * sampling loop around it (test_score.py:118-171, copied in tune_hparams_score.py:100-148).  This
 * library replaces exactly that: the host side (Python, score_based_channels_amd/) builds a *plan* -- an
 * ordered list of fused operator launches over NHWC float32 device buffers it owns -- and the library
 * executes it on a HIP stream.  One plan = one score evaluation, optionally followed by the
 * data-consistency + Langevin update + NMSE of one step, so that a whole trajectory is
 * sbc_plan_run(plan, stream, n_steps) with no host synchronisation (the reference syncs every step,
 * test_score.py:137,170).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer borrowed for the duration of the
 *     call / the lifetime of the plan (buffers stay owned by the caller, normally torch tensors);
 *   - activations are NHWC float32: x[n][h][w][c]; a complex64 [B][Nt][Nr] tensor IS a 2-channel NHWC
 *     tensor (re, im interleaved), which removes the reference's real/complex view copies
 *     (test_score.py:149,153-154);
 *   - every function returns 0 on success or a negative sbc_status; sbc_last_error() gives the message of
 *     the calling thread's last failure.  Nothing throws across the ABI.  Launches are asynchronous on the
 *     given stream (pass torch.cuda.current_stream().cuda_stream); no hidden device synchronisation
 *     except in the functions documented as synchronising.
 */
#ifndef SBC_HIP_H
#define SBC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBC_ABI_VERSION 14

typedef enum sbc_status {
    SBC_OK = 0,
    SBC_ERR_INVALID = -1,      /* bad argument / unsupported shape */
    SBC_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    SBC_ERR_UNSUPPORTED = -3   /* no kernel instantiation for this (cin, cout, ksize) */
} sbc_status;

/* Operator kinds.  Reference counterparts (file:line under /root/reference): */
typedef enum sbc_op_kind {
    SBC_OP_BEGIN_CONV = 1,   /* h = 2x-1; begin_conv 2->ngf 3x3 + bias          ncsnv2.py:270-275            */
    SBC_OP_INORM_STATS = 2,  /* InstanceNorm2dPlus statistics -> (mu, scale, shift) normalization.py:163-176 */
    SBC_OP_CONV = 3,         /* nn.Conv2d 1x1 / 3x3 / dilated 3x3 with fused prologue+epilogue  layers.py:28-60,
                                ResidualBlock :443-456, RCUBlock :126-134, CRPBlock :76-83, MSFBlock :178-184,
                                ConvMeanPool :309-313                                                        */
    SBC_OP_MAXPOOL5 = 4,     /* nn.MaxPool2d(5, stride 1, pad 2) (+ELU of the input)  layers.py:69,77-80      */
    SBC_OP_END_CONV = 5,     /* normalizer -> ELU -> end_conv ngf->2 -> / sigma   ncsnv2.py:291-298           */
    SBC_OP_LANGEVIN = 6,     /* P^H(PX-Y), noise, Langevin update, NMSE           test_score.py:156-170       */
    SBC_OP_STEP_INC = 7,     /* advance the device-side step counter (trailing_idx, test_score.py:171)        */
    SBC_OP_MEASURE = 8,      /* Y = P H + sqrt(noise) n                           test_score.py:122-124       */
    /* --- denoising-score-matching training step: ncsnv2/losses/dsm.py:6-32,
     *     train_score.py:145-173.  Reverse-mode counterparts of the operators above; see "Training operators". */
    SBC_OP_DSM_PERTURB = 9,  /* x~ = x + sigma_b z, z ~ N(0,1)                    dsm.py:14-17                */
    SBC_OP_DSM_LOSS = 10,    /* 1/2 |s + z/sigma|^2 sigma^p per sample (+ d/ds)   dsm.py:19-32                */
    SBC_OP_GRAD_ADD = 11,    /* out (+)= grad [* ELU'(in)]                        backward of +, of nn.ELU    */
    SBC_OP_INORM_BWD = 12,   /* backward of InstanceNorm2dPlus (+ ELU)            normalization.py:163-176    */
    SBC_OP_MAXPOOL5_BWD = 13,/* backward of nn.MaxPool2d(5,1,2) (+ ELU of input)  layers.py:69,77-80          */
    SBC_OP_UPSAMPLE_BWD = 14,/* adjoint of the bilinear(align_corners) resize     layers.py:182               */
    SBC_OP_POOL_BWD = 15,    /* adjoint of the 2x2 mean pool of ConvMeanPool      layers.py:311-312           */
    SBC_OP_CONV_WGRAD = 16,  /* d loss / d weight, d bias of an SBC_OP_CONV       layers.py:28-60             */
    SBC_OP_PACK_WEIGHT = 17, /* torch-layout weight (device) -> split-bf16 fragments, optionally of the adjoint conv */
    SBC_OP_END_CONV_BWD = 18,/* backward of SBC_OP_END_CONV up to the ELU output  ncsnv2.py:291-298           */
    SBC_OP_BEGIN_CONV_BWD = 19, /* weight / bias gradient of SBC_OP_BEGIN_CONV    ncsnv2.py:270-275           */
    SBC_OP_ADAM_EMA = 20,    /* torch.optim.Adam step + EMAHelper.update          losses/__init__.py:3-7, ema.py:17-22 */
    SBC_OP_CONV_PAIR = 21,   /* one RCU block in one launch: out = x + conv2(ELU(conv1(ELU(x))))   layers.py:126-134;
                                32 (or, fp16 weights, 64) channels, 3x3, no bias; the intermediate stays in LDS
                                (csrc/conv_pair.hip)                                                                  */
    SBC_OP_CONV_POOL = 22,   /* (ABI 11) one CRP stage in one launch: out = conv3x3(ELU?(MaxPool5x5(x))) [+ (res2 + ELU?(res1))]
                                layers.py:76-83; replaces an SBC_OP_MAXPOOL5 record and the SBC_OP_CONV that reads it: the
                                pooled tensor never exists in memory (csrc/conv_pair.hip: conv_pool_kernel)                  */
    SBC_OP_RES_BLOCK = 23,   /* (ABI 12) one ResidualBlock without resampling in one launch:
                                out = x + conv2(ELU(norm2(conv1(ELU(norm1(x))))))   layers.py:443-456, normalization.py:150-176;
                                32 channels, 64 x 16 samples: a workgroup owns a whole sample, so it forms the InstanceNorm++
                                statistics of the intermediate itself (csrc/conv_res.hip)                                    */
    SBC_OP_CONV_DOWN = 25,   /* (ABI 13) the tail of a downsampling ResidualBlock in one launch (layers.py:443-456 with ConvMeanPool :309-313):
                                out = meanpool2(conv3x3(ELU(norm(in))) + bias) + meanpool2(conv1x1(res1) + bias2), as a 4x4 stride-2 and a
                                2x2 stride-2 direct convolution with the pooled filters (weight_split = sbc_pack_conv_weight_pooled_f16x2 of the
                                3x3 weight, weight2_split = the same of the 1x1 shortcut weight); in = conv1's output, res1 = the block's input,
                                both [B][H][W][cin], stats = the norm's (mu, scale, shift); 32 -> 64 channels at W = 16, 64 -> 64 at W = 8,
                                H a multiple of 16; SBC_CONV_F16X2 only (csrc/conv_down.hip).  Calibration only (no kernel reads them, NULL is
                                fine): weight / weight_wino_split = the UNPOOLED direct (sbc_pack_conv_weight_f16x2) and Winograd f16x2 forms of the
                                3x3 layer, weight_wino = the unpooled direct f16x2 form of the 1x1 shortcut -- sbc_f16x2_calibrate writes the two
                                layers' activation scales into those trailers too (a host that binds the same weight buffer at a size where the
                                block is not down-fusable runs the unfused launches with the same scales)                                  */
    SBC_OP_CHAIN = 24        /* (ABI 13) a CHAIN of RCU blocks, CRP blocks and ResidualBlocks in ONE launch, for the low
                                resolution levels (8 x 2 samples of 64 or 128 channels, 16 x 4 samples of 64, 32 x 8 samples of 32 or
                                64; layers.py:76-83,126-134,234-249,443-456): a workgroup owns eight (four, one or two) samples, the
                                running tensor x stays in registers between the blocks and the convolution operands in LDS; only
                                the filters stream (csrc/conv_chain.hip).  ext = sbc_chain                                    */
} sbc_op_kind;

/* sbc_op.flags for SBC_OP_CONV / SBC_OP_MAXPOOL5 */
#define SBC_PRO_ELU      0x001  /* apply ELU to the input while staging it                                  */
#define SBC_PRO_NORM     0x002  /* apply (x - mu) * scale + shift from `stats` first (InstanceNorm++)       */
#define SBC_PRO_NORM_SELF 0x004 /* (ABI 10) with SBC_PRO_NORM, 3x3 convolutions on the matrix-core kernels (weight_split set), images of at most 64
                                   pixels (H*W a power of two): the launch computes the InstanceNorm++ statistics of its input
                                   ITSELF -- a workgroup's tile holds whole samples there -- and `stats` points at the norm's
                                   parameters [3][cin] = (alpha | gamma | beta) instead of at the output of an
                                   SBC_OP_INORM_STATS launch, which then does not exist (normalization.py:163-176)          */
#define SBC_PRO_ELU_ACC  0x20000 /* (ABI 11) with SBC_PRO_ELU: evaluate ELU with fp32's relative accuracy for small negative inputs too (a
                                   polynomial below |x| = 1/32 instead of exp(x) - 1, whose absolute error of 6e-8 is 6e-5 of an
                                   activation of -1e-3): ten vector instructions per value instead of four.  The Python host sets
                                   it in the exact modes (conv_mode 0 / 1); in conv_mode 3 the layer's weight trailer asks for it
                                   when sbc_f16x2_calibrate finds the layer's input maximum below 0.5                       */
#define SBC_EPI_RES1_ELU 0x010  /* ELU the res1 operand before adding (CRP: x = act(x))                    */
#define SBC_EPI_POOL     0x020  /* 2x2 mean pool of (conv + bias), then + res1 (ConvMeanPool)               */
#define SBC_EPI_UP       0x040  /* + bilinear(align_corners) resize of `up` [B][up_h][up_w][cout] (MSF)     */
#define SBC_EPI_ELUGRAD  0x080  /* reverse pass (split-bf16 kernels, no pool): out = conv * ELU'(res2) [+ res1]; res2 = the
                                   forward input the ELU was applied to, res1 = the gradient collected so far (may be
                                   `out` itself: every element is read and written by the same thread)              */
#define SBC_BWD_ACCUM     0x200  /* training operators: add to `out` instead of overwriting it (a tensor with several
                                   consumers collects one gradient term per consumer)                            */
#define SBC_PACK_ADJOINT  0x400  /* SBC_OP_PACK_WEIGHT: pack w'[ci][co][kh][kw] = w[co][ci][k-1-kh][k-1-kw], the weight of
                                   the adjoint (input-gradient) convolution, which then runs as an ordinary SBC_OP_CONV */
#define SBC_OP_SIDE       0x800  /* any kind, inside a plan: this launch may overlap the launches that follow it.  The plan
                                   runs it on a stream of its own that first waits for everything issued before it; side
                                   launches keep their order among themselves.  The caller guarantees that no later launch
                                   writes what it reads or reads what it writes before the next SBC_OP_JOIN             */
#define SBC_OP_JOIN       0x1000 /* this launch (and everything after it) waits for all side launches issued so far; the
                                   end of the plan always joins                                                       */
#define SBC_PACK_WINOGRAD 0x2000 /* SBC_OP_PACK_WEIGHT: the Winograd F(2x2,3x3) form (sbc_pack_conv_weight_winograd_split layout) of a
                                   3x3 weight; combines with SBC_PACK_ADJOINT                                          */
#define SBC_PRO_NORM_MOMENTS 0x4000 /* INORM_STATS: `in` holds the tensor's TILE MOMENTS [B][H*W/128][cin][2] = (mean, sum (x - mean)^2) of
                                      each 128-pixel tile, written by the launch that produced the tensor (SBC_EPI_MOMENTS_OUT);
                                      H, W, cin describe the tensor.  The statistics launch then reads a few KB per sample
                                      instead of the tensor (32 or, ABI 10, 64 channels)                                   */
#define SBC_EPI_MOMENTS_OUT 0x8000 /* CONV (Winograd split kernels, 32 or -- ABI 10 -- 64 output channels, whole 128-pixel tiles, no
                                      pool) and BEGIN_CONV: also write the tile moments of the output to `aux` [B][H*W/128][cout][2] */
#define SBC_CONV_F16W    0x100  /* fp16 weights (BASELINE config 5): `weight_split` / `weight_wino_split` hold ONE
                                   fp16 term per weight (sbc_pack_conv_weight_f16 / _winograd_f16) instead of
                                   three bf16 terms; activations are rounded to fp16 as they enter the matrix
                                   cores (v_mfma_f32_32x32x16_f16), accumulation stays fp32                    */

#define SBC_CONV_F16X2   0x10000 /* fp32-class arithmetic on the fp16 matrix cores: `weight_split` / `weight_wino_split` hold TWO
                                   fp16 terms per (scaled) weight plus a 16-byte trailer with the scales
                                   (sbc_pack_conv_weight_f16x2 / _winograd_f16x2); activations are scaled by the layer's
                                   act_scale (a power of two; sbc_f16x2_calibrate) and split into two fp16 terms as they enter
                                   the matrix cores; three fp16 MFMAs per product block (hh + hl + lh), fp32 accumulation.
                                   Representation error <= 2^-22 per operand while 2^-3 <= |x| act_scale < 16000; outside that
                                   window the device's range flag (sbc_range_flag) is raised instead of returning silently
                                   degraded numbers                                                                       */

/* One fused launch.  Unused fields are 0 / NULL.  Tensor shapes per kind:
 *   BEGIN_CONV  in [B][H][W][2], weight [cout][2][3][3] (torch layout), bias [cout], out [B][H][W][cout]
 *   INORM_STATS in [B][H][W][cin], weight = alpha|gamma|beta [3][cin], out = stats [B][3][cin]
 *   CONV        in [B][H][W][cin], weight = sbc_pack layout [k*k][cin/8][cout/32][64][4], bias [cout] or
 *               NULL, stats [B][3][cin] (PRO_NORM), res1/res2 [B][Ho][Wo][cout] or NULL, up (EPI_UP),
 *               out [B][Ho][Wo][cout] with Ho,Wo = H,W or H/2,W/2 (EPI_POOL).
 *               epilogue:  v = conv + bias;  [POOL: v = mean2x2(v)]
 *                          r = res1 [ELU];  if res2: r = res2 + r;  v = v + r;  [UP: v = v + resize(up)]
 *   MAXPOOL5    in/out [B][H][W][cin]; PRO_ELU applies ELU (monotone, so pooled after or before is equal)
 *   END_CONV    in [B][H][W][cin], stats, weight [2][cin][3][3] (torch layout), bias [2], out [B][H][W][2];
 *               divides by sigmas[labels[b]] if labels != NULL else by sigma_of_step[*step]
 *   LANGEVIN    see sbc_langevin below (passed through `ext`)
 *   CONV_PAIR   in / out [B][H][W][C] (distinct buffers), weight_split + weight2_split, flags = SBC_CONV_F16X2 or
 *               SBC_CONV_F16W; C = 32: W in {8, 16} with H % 8 == 0, or (SBC_CONV_F16W only) W in {32, 64} with H % 4 == 0;
 *               C = 64 (SBC_CONV_F16W only): W = 16 with H % 8 == 0 or W = 32 with H % 4 == 0.  The same numbers as the two CONV records it replaces
 *               (PRO_ELU; PRO_ELU + res1 = in) up to fp32 summation order.
 *   CONV_POOL   in / out [B][H][16][32] (distinct buffers), weight_split (the form of CONV_PAIR), no bias; flags = SBC_CONV_F16X2 or
 *               SBC_CONV_F16W, optionally SBC_PRO_ELU (ELU of the pooled values: pool(ELU(x)) = ELU(pool(x))) and SBC_EPI_RES1_ELU;
 *               res1 / res2 as in CONV (r = res1 [ELU]; if res2: r = res2 + r; out = conv + r); H % 8 == 0.  The same numbers as
 *               the MAXPOOL5 + CONV records it replaces up to fp32 summation order (direct instead of Winograd form).
 *   RES_BLOCK   in / out [B][64][16][32] (distinct buffers); stats = the (mu, scale, shift) table of norm1 [B][3][32] (an INORM_STATS
 *               output); weight_split / bias = conv1, weight2_split / bias2 = conv2 (sbc_pack_conv_weight_f16x2; flags =
 *               SBC_CONV_F16X2); norm2 = alpha | gamma | beta of the second norm [3][32]; with SBC_EPI_MOMENTS_OUT, aux = the output's
 *               tile moments [B][8][32][2] (what a CONV with that flag writes).  The same numbers as the CONV (PRO_NORM | PRO_ELU),
 *               INORM_STATS, CONV (PRO_NORM | PRO_ELU, res1 = in) records it replaces up to fp32 summation order.
 */
typedef struct sbc_op {
    int32_t kind, flags;
    int32_t B, H, W;
    int32_t cin, cout, ksize, dil;
    int32_t up_h, up_w;
    int32_t tag;                 /* free label; sbc_plan_profile times all ops carrying a given tag.  tag 1 = the 3x3 ngf -> ngf
                                    layers at full resolution: they also run under their own kernel symbol so that profilers
                                    report them separately */
    const void* in;
    void* out;
    const void* weight;
    const void* bias;
    const void* stats;
    const void* res1;
    const void* res2;
    const void* up;
    const void* ext;             /* kind-specific extension struct (sbc_langevin / sbc_endconv), host memory,
                                    copied at sbc_plan_create / read during sbc_op_launch */
    const void* weight_wino;     /* CONV, optional: the same 3x3 weight in Winograd F(2x2,3x3) form,
                                    sbc_pack_conv_weight_winograd layout [16][cin/8][cout/32][64][4]; used for
                                    undilated 3x3 convolutions on power-of-two images (2.25x fewer multiplies),
                                    `weight` stays the fallback for every other shape */
    const void* weight_split;    /* CONV, optional: the same weight with every fp32 value split exactly into three
                                    bf16 terms, sbc_pack_conv_weight_split layout [k*k][cin/16][cout/32][3][64][8]
                                    (uint16).  When set, the convolution runs on the bf16 matrix cores as six bf16
                                    MFMAs per fp32 product block with fp32 accumulation (fp32-level accuracy, see
                                    csrc/conv_x3.hip); takes precedence over `weight_wino` and `weight`, which then
                                    may be NULL.  SBC_CONV_MODE=f32 in the environment ignores it. */
    const void* weight_wino_split; /* CONV, optional: the Winograd form of a 3x3 weight with every value split into three
                                    bf16 terms, sbc_pack_conv_weight_winograd_split layout [16][cin/16][cout/32][3][64][8]
                                    (uint16): Winograd F(2x2,3x3) with its 16 products on the bf16 matrix cores
                                    (csrc/conv_wx3.hip).  Used for undilated 3x3 convolutions on power-of-two images,
                                    ahead of `weight_split`. */
    /* --- training operators only (ABI 7; NULL / unused on the inference path) --- */
    const void* grad;            /* incoming gradient: d loss / d (this operator's forward output) */
    void* aux;                   /* kind-specific second output or scratch (see "Training operators") */
    void* wgrad;                 /* parameter-gradient output: conv weight in torch layout, or alpha|gamma|beta [3][cin] */
    void* bgrad;                 /* bias-gradient output [cout] or NULL */
    /* --- ABI 9 --- */
    const void* weight2_split;   /* CONV_PAIR: the second convolution's weight in the form `weight_split` holds the first one's
                                    (sbc_pack_conv_weight_f16x2 with SBC_CONV_F16X2, sbc_pack_conv_weight_f16 with SBC_CONV_F16W) */
    /* --- ABI 11 --- */
    void* calib;                 /* NULL.  (Set by sbc_f16x2_calibrate on its private copies of the records: two device floats per
                                    f16x2 convolution that collect max |x| of what the launch stages.) */
    /* --- ABI 12 --- */
    const void* bias2;           /* RES_BLOCK: bias of the second convolution [cout] */
    const void* norm2;           /* RES_BLOCK: alpha | gamma | beta of the second InstanceNorm++ [3][cout] */
    /* --- ABI 13 --- */
    const void* weight2_wino_split; /* CONV_PAIR / RES_BLOCK: the second convolution's Winograd f16x2 form, or NULL.  No kernel of a fused
                                    record reads it (nor `weight_wino_split` there): sbc_f16x2_calibrate writes the layer's activation
                                    scale into the trailer of EVERY form it is handed, so that a host which shares one weight buffer
                                    between array sizes -- fused at one, unfused Winograd at another -- finds the same scale in both */
    /* --- ABI 14: launch lanes of a plan (all zero: the record runs on the run stream in list order, as before) --- */
    int32_t lane;                /* 0 = the stream handed to sbc_plan_run; 1 .. SBC_MAX_LANES-1 = a stream of the library that goes with that
                                    run stream (probed once to sit on another hardware queue; shared by the plans run on it).  Records of one
                                    lane run in list order; records of different lanes are ordered ONLY by the events below (and by the
                                    start and the end of an sbc_plan_run call, which every lane is forked from / joined into) */
    int32_t signal;              /* 0, or an event id 1 .. SBC_MAX_EVENTS: recorded on this record's lane right behind it */
    int32_t wait[2];             /* 0, or event ids this record's lane waits for in front of it; the id must be signalled by a record
                                    EARLIER in the list (of the same iteration).  The host that builds the list owns the hazards: a
                                    record must not write what a concurrently running lane reads or writes (plan.assign_slots) */
} sbc_op;
#define SBC_MAX_LANES 4
#define SBC_MAX_EVENTS 64

/* Training operators.  The reverse of a forward record `y = epi(conv(pro(x)))` is built by the host
 * (score_based_channels_amd/train.py) from these pieces; every tensor is NHWC float32 like its forward counterpart, `grad`
 * always has the shape of the forward OUTPUT of the operator being reversed, `out` the shape of its forward INPUT.
 *   DSM_PERTURB   in = samples [B][n] (n = H*W*cin), ext = sbc_dsm; out = samples + sigmas[labels[b]] * z;
 *                 aux = the scaled noise sigma_b * z [B][n] (kept for the loss).  z = ext.noise (standard normal draws,
 *                 replay) or in-kernel Philox keyed by (seed, sample_id[b] or b, offset, element).
 *   DSM_LOSS      in = scores [B][n], grad = scaled noise [B][n], ext = sbc_dsm; out = per-sample loss [B]:
 *                 1/2 * sum_n (s - t)^2 * sigma^p with t = -1/sigma^2 * noise (dsm.py:20-30; the batch mean of :32 is the
 *                 caller's); aux (optional) = d mean-loss / d scores = (s - t) * sigma^p / B.
 *   GRAD_ADD      out (+)= grad, or grad * ELU'(in) with SBC_PRO_ELU (in = the tensor the forward ELU was applied to).
 *   INORM_BWD     reverse of stats -> affine -> ELU: in = x, stats [B][3][cin] (forward), weight = alpha|gamma|beta,
 *                 grad = d / d ELU-output (or d / d norm-output without SBC_PRO_ELU); out (+)= d / d x through the
 *                 normalised value, the spatial mean / variance and the cross-channel mean term; wgrad = d alpha|gamma|beta
 *                 [3][cin] summed over the batch (written); aux = scratch, >= B*6*cin floats.
 *   MAXPOOL5_BWD  in = forward input, grad = d / d output; out (+)= the gradient routed to each window's first maximum in
 *                 row-major order (times ELU'(in) with SBC_PRO_ELU); aux = scratch, >= B*H*W*cin bytes.
 *   UPSAMPLE_BWD  grad = [B][H][W][cin]; out (+)= [B][up_h][up_w][cin], the transpose of the resize of SBC_EPI_UP.
 *   POOL_BWD      grad = [B][H/2][W/2][cin]; out = [B][H][W][cin] = grad[h/2][w/2] / 4.
 *   CONV_WGRAD    in/stats/flags(PRO_*)/ksize/dil as in the forward CONV, grad = d / d (conv + bias) [B][H][W][cout];
 *                 wgrad = [cout][cin][k][k] (torch layout, written), bgrad = [cout] or NULL; aux = scratch (float),
 *                 >= sbc_wgrad_scratch_floats(B, H, W, cin, cout, ksize) (one buffer serves all launches of a stream).
 *   PACK_WEIGHT   in = [cout][cin][k][k] float32 DEVICE, out = the sbc_pack_conv_weight_split layout (of the adjoint
 *                 convolution cout -> cin with SBC_PACK_ADJOINT).  Batched form (aux != NULL): aux = device int32 table
 *                 [B][6] = (source offset from `in` in floats, destination offset from `out` in uint16, cout, cin, k*k
 *                 or 16 for the Winograd form, adjoint) and cin/cout = those of the largest entry: every weight of a
 *                 network in one launch.
 *   END_CONV_BWD  in/stats/weight/ext as in END_CONV, grad = d / d score [B][H][W][2]; out = d / d ELU-output
 *                 [B][H][W][cin] (written); wgrad [2][cin][3][3], bgrad [2]; aux = scratch >= sbc_wgrad_scratch_floats.
 *   BEGIN_CONV_BWD in = x [B][H][W][2], grad = d / d output [B][H][W][cout]; wgrad [cout][2][3][3], bgrad [cout]; aux.
 *   ADAM_EMA      ext = sbc_adam; in = gradients [n], out = parameters [n] (updated in place), aux = state [3][n]:
 *                 exp_avg | exp_avg_sq | EMA shadow.
 */
typedef struct sbc_dsm {
    const float* sigmas;         /* [num_classes] device */
    const int64_t* labels;       /* [B] device: noise level of each sample (dsm.py:9-12) */
    const float* noise;          /* [B][n] device standard-normal draws to replay, or NULL -> Philox */
    const int64_t* sample_id;    /* [B] device Philox stream ids or NULL (= b) */
    uint64_t seed;
    int32_t offset;              /* Philox counter word 1 (the optimiser step) = offset + *step */
    float anneal_power;          /* dsm.py:7 (2 in train_score.py:55) */
    const int32_t* step;         /* device step counter or NULL (= 0): lets a replayed plan draw fresh noise every step */
    float grad_scale;            /* DSM_LOSS: extra factor on d loss / d scores; 0 means 1.  1 / world_size makes the SUM
                                    all-reduce of data-parallel ranks the gradient of the mean over the global batch */
} sbc_dsm;

typedef struct sbc_adam {
    int64_t n;                   /* elements */
    double lr, beta1, beta2, eps; /* torch.optim.Adam(lr, betas, eps), weight_decay = 0, amsgrad = False.  double: torch
                                    forms 1 - beta and the bias corrections from python floats before rounding to fp32 */
    double ema_mu;               /* EMAHelper(mu): shadow = (1 - mu) * p + mu * shadow; < 0 disables the shadow update */
    const int32_t* step;         /* device counter: number of optimiser steps already taken (t - 1) */
} sbc_adam;

/* Extension of SBC_OP_CHAIN: the blocks, in execution order.  Every convolution is 3x3, cin = cout = op.cin:
 *   SBC_CHAIN_RCU   x <- x + conv_w2(ELU(conv_w1(ELU(x))))                      (no bias)                 layers.py:126-134
 *   SBC_CHAIN_CRP   x <- ELU(x); p = conv_w1(maxpool5(x)); x <- p + x; q = conv_w2(maxpool5(p)); x <- q + x   (no bias)   layers.py:76-83
 *   SBC_CHAIN_RES   a ResidualBlock without resampling or channel change (layers.py:443-456, normalization.py:163-176):
 *                   x <- shortcut + conv_w2(ELU(norm2(conv_w1(ELU(norm1(x))) + bias1))) + bias2, every convolution with dilation
 *                   dil (1, or 2 / 4 at a width of two); shortcut = x, or conv_w3(x) + bias3 when w3 is set (the dilated 'down'
 *                   blocks); norm1 / norm2 = alpha | gamma | beta [3][cin] of the two InstanceNorm++ layers: the launch forms their
 *                   statistics itself, a workgroup holds whole samples
 * w1 / w2 / w3: sbc_pack_conv_weight_f16x2 forms (SBC_CONV_F16X2 must be set: the only multiplier the kernel has).  w1_wino / w2_wino:
 * the same layers' sbc_pack_conv_weight_winograd_f16x2 forms or NULL -- never read by a kernel, but sbc_f16x2_calibrate writes the
 * layer's activation scale into every form it is handed (see sbc_op.weight2_wino_split). */
#define SBC_CHAIN_MAX_BLOCKS 6
#define SBC_CHAIN_RCU 0
#define SBC_CHAIN_CRP 1
#define SBC_CHAIN_RES 2
typedef struct sbc_chain {
    int32_t n_blocks;
    int32_t type[SBC_CHAIN_MAX_BLOCKS];
    int32_t dil[SBC_CHAIN_MAX_BLOCKS];             /* RES blocks; 0 or 1 = undilated */
    const void* w1[SBC_CHAIN_MAX_BLOCKS];
    const void* w2[SBC_CHAIN_MAX_BLOCKS];
    const void* w1_wino[SBC_CHAIN_MAX_BLOCKS];
    const void* w2_wino[SBC_CHAIN_MAX_BLOCKS];
    const void* w3[SBC_CHAIN_MAX_BLOCKS];          /* RES: shortcut convolution or NULL */
    const float* bias1[SBC_CHAIN_MAX_BLOCKS];      /* RES */
    const float* bias2[SBC_CHAIN_MAX_BLOCKS];
    const float* bias3[SBC_CHAIN_MAX_BLOCKS];
    const float* norm1[SBC_CHAIN_MAX_BLOCKS];
    const float* norm2[SBC_CHAIN_MAX_BLOCKS];
} sbc_chain;

/* Extension of SBC_OP_END_CONV: where the noise level comes from (ncsnv2.py:295-298). */
typedef struct sbc_endconv {
    const float* sigmas;         /* [num_classes] device, float32 (models/__init__.py:4-8) */
    const int64_t* labels;       /* [B] device or NULL */
    const float* sigma_of_step;  /* [n_steps] device: sigma of the level walked at step k (if labels NULL) */
    const int32_t* step;         /* device step counter */
} sbc_endconv;

/* Extension of SBC_OP_LANGEVIN and SBC_OP_MEASURE.  All complex tensors are interleaved float32 pairs.
 *   X      [B][Nt][Nr]  current estimate, updated in place            (test_score.py:164-165)
 *   score  [B][Nt][Nr]  output of the score plan                       (:150-154)
 *   P      [nP][Np][Nt] conj-transposed pilots, sample b uses P[p_index[b]] (b if p_index NULL)   (:109-111)
 *   Y      [B][Np][Nr]  measurements                                   (:122-124)
 *   Htrue  [nH][Nt][Nr] ground truth, sample b uses Htrue[h_index[b]]  (:112-113,131)
 *   sched  [G][n_steps][4] float32 (alpha, dc_div, noise_scale, dc_boost) of group `group[b]` at step k,
 *          i.e. the float64 python scalars of :143-144,160,165 rounded to float32; dc_boost (1 for test_score,
 *          --dc_boost of test_mmse.py:231-233) multiplies the data-consistency gradient before the division; 0 means
 *          "not set" and is read as 1 (hosts written against the three-column table of ABI <= 4)
 *   noise  [n_steps][B][Nt][Nr] CN(0,1) draws or NULL -> in-kernel Philox4x32-10: elements 2q, 2q+1 of a trajectory share
 *          the block of counter (q, step, traj_id lo, traj_id hi), key = seed; words (x, y) / (z, w) feed one Box-Muller
 *          each (csrc/philox.h; restated on the host by oracle/ald_oracle.py::device_complex_normal)      (:160-161)
 *   nmse   [n_steps][B] float32 log, row *step is written               (:168-170)
 * SBC_OP_MEASURE reads Htrue and P and writes Y; its `noise` is [B][Np][Nr] and `meas_scale[b]` =
 * float32(sqrt(local_noise)).
 * Constraints of SBC_OP_LANGEVIN (checked, SBC_ERR_INVALID otherwise): Nr is EVEN -- the update takes the elements of a row in
 * adjacent pairs (one Philox block, one pilot value, 16-byte accesses) -- and X, score, Y, Htrue and noise are 16-byte aligned
 * (Nt * Nr even keeps every per-trajectory slice aligned too).  The score network itself needs Nt, Nr multiples of 8.
 */
typedef struct sbc_langevin {
    float* X;
    const float* score;
    const float* P;
    const int32_t* p_index;
    float* Y;
    const float* Htrue;
    const int32_t* h_index;
    const float* sched;
    const int32_t* group;
    const float* noise;
    float* nmse;
    const int32_t* step;
    const int64_t* traj_id;
    const float* meas_scale;
    uint64_t seed;
    int32_t n_steps, Nt, Nr, Np;
} sbc_langevin;

typedef struct sbc_plan sbc_plan;

/* --- library ----------------------------------------------------------------------------------- */
int sbc_abi_version(void);
/* (ABI 12) The persistent kernels (SBC_OP_CONV_PAIR / CONV_POOL / RES_BLOCK and the direct kernel behind SBC_OP_CONV) size their grids for
 * `n` CUs instead of all of them; 0 = all (the default).  A host that keeps TWO independent launch streams busy sets half the device's
 * CUs, so that the two streams' persistent launches are resident side by side instead of one after the other (measured: -0.7 % per
 * two-stream Langevin step; results do not depend on it).  Process-wide DEFAULT; takes effect for launches issued after the call.
 * Prefer sbc_plan_set_persistent_cus (below): a plan's own width does not race with other threads, handles or devices. */
int sbc_set_persistent_cus(int32_t n);
const char* sbc_last_error(void);
/* number of visible HIP devices, or a negative sbc_status (does not initialise a device context) */
int sbc_device_count(void);

/* --- single operator (unit tests, ad-hoc use) ---------------------------------------------------- */
/* Launch one fused operator asynchronously on `stream` (a hipStream_t, may be NULL = default stream). */
int sbc_op_launch(const sbc_op* op, void* stream);

/* --- plans ---------------------------------------------------------------------------------------
 * sbc_plan_create copies `ops` (and their `ext` structs); the device buffers they point to must outlive the
 * plan.  sbc_plan_run executes the whole op list `n_iters` times in order on `stream` -- records with a `lane` (ABI 14) on the
 * library's lane streams: every lane is forked from `stream` at the start of the call and joined into it at its end (not between
 * the iterations, where the records' own events order the lanes), so that a caller sees ONE asynchronous unit of work.  With
 * use_graph = 1 and lanes the captured graph is flat (lane records on the run stream in list order); use_graph = 2 keeps the lanes as
 * parallel branches of the graph (experimental: the runtime's graph launch is not robust with them, see csrc/api.hip).  With
 * use_graph != 0 the op list is captured once into a hipGraph (on first use for that stream) and replayed;
 * this is legal because nothing in a plan depends on host state -- step-dependent scalars are read from
 * device tables through the device step counter. */
int sbc_plan_create(const sbc_op* ops, int32_t n_ops, sbc_plan** out_plan);
int sbc_plan_run(sbc_plan* plan, void* stream, int32_t n_iters, int32_t use_graph);
/* (ABI 13) Grid width, in CUs, of the persistent kernels of THIS plan's launches (0 = the process default above).  The setting is a
 * field of the plan, applied on the calling host thread for the duration of sbc_plan_run (and baked into a captured graph; changing
 * it drops the captured graph): plans driven from different host threads, for different streams or devices, are independent of each
 * other -- the library holds no mutable state shared between handles on this path.  Results never depend on it. */
int sbc_plan_set_persistent_cus(sbc_plan* plan, int32_t n);
void sbc_plan_destroy(sbc_plan* plan);

/* Per-kernel timing for the roofline report: while enabled (tag >= 0), every launch of an op whose
 * `tag` matches is bracketed by a hipEvent pair on the launch stream (eager runs only).  sbc_plan_profile_read
 * synchronises on the recorded events, returns the summed kernel time and launch count since enabling, and
 * resets the accumulators. */
int sbc_plan_profile(sbc_plan* plan, int32_t tag);
int sbc_plan_profile_read(sbc_plan* plan, double* total_ms, int64_t* n_launches);

/* --- helpers --------------------------------------------------------------------------------------
 * Host-side re-ordering of a torch-layout convolution weight [cout][cin][k][k] into the MFMA B-operand
 * fragment order consumed by SBC_OP_CONV: [k*k][cin/8][cout/32][64 lanes][4].  dst and src are HOST
 * pointers; cin % 8 == 0, cout % 32 == 0. */
int sbc_pack_conv_weight(const float* src, int32_t cout, int32_t cin, int32_t ksize, float* dst);
/* Same for the Winograd form of a 3x3 weight: U = G g G^T per (cout, cin) (computed in double), packed with the 16
 * transform positions in place of the taps: [16][cin/8][cout/32][64][4]. */
int sbc_pack_conv_weight_winograd(const float* src, int32_t cout, int32_t cin, float* dst);
/* Split-bf16 form: w = wh + wm + wl with wh = bf16(w), wm = bf16(w - wh), wl = bf16(w - wh - wm) (round to nearest
 * even; the sum is exact), as bf16 bit patterns in B-operand fragment order [k*k][cin/16][cout/32][3][64 lanes][8]:
 * lane l of block (tap, g, n) holds w[n*32 + (l & 31)][g*16 + 8*(l >> 5) + j][tap], j = 0..7.  cin % 16 == 0. */
int sbc_pack_conv_weight_split(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst);
/* Winograd form U = G g G^T (double, rounded once to float) of a 3x3 weight, split like sbc_pack_conv_weight_split with
 * the 16 transform positions in place of the taps: [16][cin/16][cout/32][3][64][8] uint16. */
int sbc_pack_conv_weight_winograd_split(const float* src, int32_t cout, int32_t cin, uint16_t* dst);

/* --- the whole score network behind one call (hosts that are not Python) -------------------------------
 * The Python host (score_based_channels_amd/plan.py, scorenet.py) wires NCSNv2Deepest.forward
 * (ncsnv2/models/ncsnv2.py:269-300) into ~150 sbc_op records, shares activation storage, packs the weights and binds
 * device memory.  sbc_score_create does all of that inside the library from the tensors of the reference checkpoint's
 * `model_state` (names as in NCSNv2Deepest.state_dict(): "res2.0.conv2.conv.weight", "normalizer.alpha", ...; HOST
 * pointers, float32, torch layouts).  The handle owns its device memory (weights, activation slots for `batch` samples,
 * labels).  conv_mode: 0 = split-bf16 (fp32-accurate), 1 = fp32 MFMA, 2 = fp16 weights (SBC_CONV_F16W; every parameter is
 * rounded to fp16 first), 3 = f16x2 (SBC_CONV_F16X2: fp32-class on the fp16 matrix cores, default of the Python host).
 *   sbc_score_buffers      device pointers of the input x [batch][Nt][Nr][2], the output score (same shape) and the
 *                          int64 noise-level labels [batch] (ncsnv2.py:295-298); fill x / labels, then
 *   sbc_score_forward      one score evaluation, asynchronous on `stream`;
 *   sbc_score_level_source inside an annealed-Langevin plan: take sigma from sigma_of_step[*step] (device) instead of
 *                          the labels (both NULL switches back);
 *   sbc_score_ops          the bound records, e.g. to append SBC_OP_LANGEVIN + SBC_OP_STEP_INC (X = the x buffer,
 *                          score = the out buffer) and build a full Langevin-step plan with sbc_plan_create. */
typedef struct sbc_tensor_ref { const char* name; const float* data; int64_t numel; } sbc_tensor_ref;
typedef struct sbc_score_desc {
    int32_t ngf, channels;       /* 32, 2 (train_score.py:37,59) */
    int32_t nt, nr;              /* array size; multiples of 8 */
    int32_t batch;
    int32_t conv_mode;
    const float* sigmas;         /* HOST [num_classes] (models/__init__.py:4-8) */
    int32_t num_classes;
    int32_t flags;               /* ABI 9: SBC_SCORE_* */
} sbc_score_desc;
#define SBC_SCORE_FOLD_STATS 0x2 /* InstanceNorm++ statistics of the full-resolution tensors from the tile moments their producers write
                                    (SBC_EPI_MOMENTS_OUT / SBC_PRO_NORM_MOMENTS; not conv_mode 1; scorenet.DEFAULT_FOLD_STATS) */
#define SBC_SCORE_FUSE_PAIRS 0x1 /* every RCU block of 32 channels at a width of 16 as one SBC_OP_CONV_PAIR record, and (ABI 11) every CRP
                                    stage of that shape as one SBC_OP_CONV_POOL record (conv_mode 2 / 3); what the Python host does
                                    by default in those modes (scorenet.DEFAULT_FUSE_PAIRS) */
#define SBC_SCORE_FUSE_RES   0x4 /* (ABI 12) the ResidualBlocks without resampling at 64 x 16 as one SBC_OP_RES_BLOCK record each (conv_mode 3);
                                    what the Python host does by default next to SBC_SCORE_FUSE_PAIRS (scorenet.DEFAULT_FUSE_RES) */
#define SBC_SCORE_FUSE_CHAIN 0x8 /* (ABI 13) the RCU / CRP runs of the 8 x 2 level as SBC_OP_CHAIN records (conv_mode 3); what the Python host does by
                                    default next to SBC_SCORE_FUSE_PAIRS (scorenet.DEFAULT_FUSE_CHAIN) */
#define SBC_SCORE_FUSE_DOWN 0x10 /* (ABI 13) pooled conv2 + pooled 1x1 shortcut of the downsampling ResidualBlocks res2.0 / res3.0 as one SBC_OP_CONV_DOWN
                                    record (conv_mode 3); what the Python host does by default (scorenet.DEFAULT_FUSE_DOWN) */
#define SBC_SCORE_FUSE_END  0x20 /* (ABI 13) the normalizer's statistics inside the SBC_OP_END_CONV launch (SBC_PRO_NORM_SELF on that record: 32 channels,
                                    1024 pixels; any conv_mode); what the Python host does by default next to SBC_SCORE_FUSE_PAIRS
                                    (scorenet.DEFAULT_FUSE_END) */
#define SBC_SCORE_SKIP_LANES 0x40 /* (ABI 14) the decoder's skip branches (refineK.adapt_convs.0) behind their anchor records on launch lane 1 (sbc_op.lane /
                                    signal / wait): for small batches, where the low-resolution launches are latency-bound -- what the Python host does for
                                    batches of at most scorenet.SKIP_OVERLAP_MAX_T trajectories (plan.hoist_skip_branches; identical results) */
typedef struct sbc_score sbc_score;
int sbc_score_create(const sbc_score_desc* desc, const sbc_tensor_ref* tensors, int32_t n_tensors, sbc_score** out);
int sbc_score_buffers(sbc_score* score, float** x, float** out, int64_t** labels);
int sbc_score_ops(sbc_score* score, const sbc_op** ops, int32_t* n_ops);
int sbc_score_level_source(sbc_score* score, const float* sigma_of_step, const int32_t* step);
int sbc_score_forward(sbc_score* score, void* stream);
void sbc_score_destroy(sbc_score* score);

/* fp16 weight forms for SBC_CONV_F16W (round to nearest even): the layouts of sbc_pack_conv_weight_split /
 * sbc_pack_conv_weight_winograd_split with a single fp16 term, [k*k | 16][cin/16][cout/32][64 lanes][8] uint16.  The
 * Winograd form rounds U = G g G^T (double) once to fp16. */
int sbc_pack_conv_weight_f16(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst);
int sbc_pack_conv_weight_winograd_f16(const float* src, int32_t cout, int32_t cin, uint16_t* dst);

/* f16x2 weight forms for SBC_CONV_F16X2: every weight (or Winograd-transformed weight U = G g G^T, double -> float) is
 * scaled by 2^s -- s chosen per layer so that the largest magnitude lies in [2^13, 2^14) -- and written as two fp16 terms
 * h = fp16(w 2^s), l = fp16(w 2^s - h), in the layout of sbc_pack_conv_weight_split with 2 terms,
 * [k*k | 16][cin/16][cout/32][2][64 lanes][8] uint16, followed by a 16-byte trailer of four float32: (act_scale = 1, descale =
 * 1 / (act_scale 2^s), 2^-s, 0); act_scale must be a power of two (sbc_f16x2_calibrate sets it per layer on the device copy; a
 * host that knows its activations may write the first two words itself).  dst holds sbc_f16x2_elems(...) uint16. */
#define SBC_F16X2_ACT_SHIFT 0
#define sbc_f16x2_elems(taps, cin, cout) ((size_t)(taps) * (cin) * (cout) * 2 + 8)
/* (ABI 13) The filter of `meanpool2(conv(x))` as ONE stride-2 convolution, in the sbc_pack_conv_weight_f16x2 form with ksize + 1 taps per side:
 * src [cout][cin][ksize][ksize] (ksize 3 -> a 4x4 filter, ksize 1 -> 2x2), W'[p][q] = 1/4 sum_{a,b in {0,1}} W[p - a][q - b] formed in double.
 * dst: sbc_f16x2_elems((ksize + 1)^2, cin, cout) uint16. */
int sbc_pack_conv_weight_pooled_f16x2(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst);
int sbc_pack_conv_weight_f16x2(const float* src, int32_t cout, int32_t cin, int32_t ksize, uint16_t* dst);
int sbc_pack_conv_weight_winograd_f16x2(const float* src, int32_t cout, int32_t cin, uint16_t* dst);
/* Range flag of the f16x2 kernels on the CURRENT device, a bit set collected since the last reset:
 *   bit 0 (SBC_RANGE_OVERFLOW)   a convolution staged an activation with |x| * act_scale >= 16000: its high fp16 term (or a
 *                                Winograd transform sum of four) may have overflowed;
 *   bit 1 (SBC_RANGE_UNDERFLOW)  a whole wavefront's share of an input tile (>= 8 pixels x every channel) was non-zero but below
 *                                2^-6 after scaling: the low fp16 terms of that region are denormal, the products there carry a
 *                                relative error above 2^-19 instead of 2^-22.
 * Either way the results of that run are not fp32-class: run the batch again in split-bf16 mode (conv_mode 0), which has fp32's
 * range -- the Python host does that by itself (driver.run_trajectories) and records it in the result file.  Synchronises with
 * the whole device (every stream).  reset != 0 clears the word. */
#define SBC_RANGE_OVERFLOW  1
#define SBC_RANGE_UNDERFLOW 2
#define SBC_RANGE_ELU       4   /* a fused RCU launch (SBC_OP_CONV_PAIR evaluates ELU as exp(x) - 1 only) ran on a layer whose calibrated
                                   inputs are below 2^-4, where that form no longer has fp32's relative accuracy */
int sbc_range_flag(int32_t* flag, int32_t reset);

/* Per-layer activation scales of conv_mode f16x2 (ABI 11).  A two-term fp16 split x s = h + l is fp32-class only while l stays a
 * normal fp16 number, i.e. for |x s| >= 2^-3; the packers write act_scale s = 1, which is right for O(1) activations only.
 * sbc_f16x2_calibrate runs the records `ops` ONCE on the first sample of their buffers (B = 1; the caller has put
 * sbc_f16x2_calibration_input there: a fixed CN(0,1)-like pattern, so the result depends on the checkpoint and the array size,
 * never on the data or the batch), collects max |x| of what every SBC_CONV_F16X2 convolution stages, and rewrites the 16-byte
 * trailers of their weight forms in DEVICE memory: act_scale = the power of two that puts that maximum into [2^8, 2^9) (31x
 * head room below the overflow guard, two-term precision for everything down to 2^-11 of the maximum), descale accordingly.
 * InstanceNorm++ makes the network's activations independent of the input's magnitude, so one calibration per checkpoint holds
 * for every noise level; data that leaves the window anyway raises sbc_range_flag.  Synchronises `stream`; must not run while
 * other launches use the same weights.  sbc_score_create (conv_mode 3) calls it; the Python host does on its first bind. */
int sbc_f16x2_calibration_input(float* x_host, int64_t n);
int sbc_f16x2_calibrate(const sbc_op* ops, int32_t n_ops, void* stream);

/* Known-answer hooks of the in-kernel random numbers (tests; synchronous, current device):
 *   sbc_debug_philox4x32     n blocks of Philox4x32-10 from host (c0, c1, c2, c3, k0, k1) records -> host (x, y, z, w) records:
 *                            the Random123 known-answer vectors must come back;
 *   sbc_debug_complex_normal the CN(0,1) draws (re, im interleaved) SBC_OP_LANGEVIN (step >= 0) / SBC_OP_MEASURE (step = -1) use
 *                            for elements 0 .. n_elem-1 of trajectory `traj` at `step` under `seed`. */
int sbc_debug_philox4x32(const uint32_t* counters_keys, int32_t n, uint32_t* out);
int sbc_debug_complex_normal(uint64_t seed, int64_t traj, int32_t step, int32_t n_elem, float* out);

/* scratch floats SBC_OP_CONV_WGRAD / END_CONV_BWD / BEGIN_CONV_BWD need in `aux` for this shape */
int64_t sbc_wgrad_scratch_floats(int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t ksize);

/* ---- Environment variables -------------------------------------------------------------------------------------------------
 * Everything the library (csrc/) and the Python host (score_based_channels_amd/) read from the environment, in ONE place.  None of them
 * changes a result: they select between launch plans / kernel variants that compute the same sums (A/B timing aids, each verified
 * bit-identical or within the stated tolerance by tests/), or configure the process (library path, distributed backend).  The
 * torchrun variables RANK / WORLD_SIZE / LOCAL_RANK are read by shard.py.  tests/test_host_logic.py::test_documented_environment_variables
 * holds this list to the getenv / os.environ sites of the tree.
 *
 *   process:    SBC_LIB_PATH (another build of this library), SBC_DIST_BACKEND (nccl | gloo; gloo lets several ranks share one GPU),
 *               SBC_DIST_TIMEOUT_S (bench.py: process-group timeout), SBC_CPU_BASELINE_WORKERS (bench.py: worker count of the cpu_baseline leg)
 *   streams:    SBC_PERSIST_CUS (grid width of the persistent kernels, overrides sbc_plan_set_persistent_cus), SBC_NO_BALANCED_GRID,
 *               SBC_STREAM_SMALL_PX, SBC_STREAM_LAG_MIN_STEPS, SBC_NO_STREAM_LAG, SBC_LAG_RECORDS (driver.run_concurrently / ald.py),
 *               SBC_NO_SKIP_OVERLAP, SBC_SKIP_OVERLAP_MAX_T (scorenet.py: the small-batch plan with the skip branches on a side stream)
 *   plan:       SBC_NO_CONV_DOWN, SBC_NO_CHAIN, SBC_NO_CHAIN4, SBC_NO_CHAIN8, SBC_NO_CHAIN8_CRP, SBC_CHAIN8_RES, SBC_NO_END_SELF,
 *               SBC_NO_RES_BLOCK, SBC_NO_CONV_POOL (plan.py: the unfused record sequence instead of the named fused record),
 *               SBC_NO_CALIB (scorenet.py: f16x2 activation scales stay 1)
 *   kernels:    SBC_NO_PAIR_P3, SBC_NO_PAIR_ROLL, SBC_PAIR_ROLL_MIN_TILES (conv_pair.hip), SBC_DP_WGS, SBC_NO_CONV_DP, SBC_NO_CONV_DP32, SBC_NO_CONV_DP_NORM
 *               (conv_dp.hip), SBC_TILE (conv_x3.hip, conv_mfma.hip), SBC_WX3_MB2 (conv_wx3.hip), SBC_WINO_MB1, SBC_NO_WINO (conv_wino.hip,
 *               conv_mfma.hip), SBC_CONV_MODE=f32, SBC_NO_WX3 (conv_mfma.hip), SBC_CHAIN_NW8, SBC_CHAIN_GD (conv_chain.hip)
 */

#ifdef __cplusplus
}
#endif
#endif /* SBC_HIP_H */
