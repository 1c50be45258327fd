"""Model family of the plugin surface: ImgGrnd -> VidGrnd (+obj_tx) -> VOGNet
(+mul_tx), each in SEP / TEMP / SPAT flavour — the nine class names the
reference selector hands out (code/mdl_vog.py:29-756, code/mdl_selector.py).

Which stages run is decided inside libvog_hip from the model descriptor
(csrc/forward.hip `build_steps`); the classes here differ only in which
parameter groups they own (= which state-dict keys exist).
"""
from __future__ import annotations

from .mdl_base import AnetBaseMdl
from .mdl_conc import ConcSEP, ConcSPAT, ConcTEMP

_LANG = ("lstm_encoder.", "lstm_out_feat_proj.", "srl_arg_words_out_enc.", "srl_simple_lin.")
_VIS = ("prop_encoder.", "seg_encoder.", "seg_verb_classf.")
_CONC = ("lin2.", "lin_tmp.")


class ImgGrnd(AnetBaseMdl):
    """Language encode + prop/seg encoders + vis||lang -> lin2 scoring
    (reference mdl_vog.py:29-397)."""
    kind = "igrnd"

    def set_args_mdl(self):
        m = self.cfg.mdl
        self.prop_dim = m.prop_feat_dim
        self.prop_encode_dim = m.vsrl.prop_encode_size
        self.seg_feat_dim = m.seg_feat_dim
        self.seg_feat_encode_dim = m.vsrl.seg_encode_size
        self.lang_encode_dim = m.vsrl.lang_encode_size
        self.prop_seg_feat_dim = self.prop_encode_dim + self.seg_feat_encode_dim
        self.vis_lang_feat_dim = self.prop_seg_feat_dim + self.lang_encode_dim
        assert m.name == self.kind, f"cfg.mdl.name={m.name!r} but class is {self.kind!r}"

    def build_lang_model(self):
        self._take(_LANG)

    def build_vis_model(self):
        self._take(_VIS)

    def build_conc_model(self):
        self._take(_CONC)


class VidGrnd(ImgGrnd):
    """+ object transformer over the proposals of a video (mdl_vog.py:412-523)."""
    kind = "vgrnd"

    def build_vis_model(self):
        ImgGrnd.build_vis_model(self)
        self._take(("obj_txf.", "pe_obj_sub_enc."))
        self.vid_w = self.cfg.ds.resized_width
        self.vid_h = self.cfg.ds.resized_height


class VOGNet(VidGrnd):
    """+ multimodal transformer over (argument, proposal) tokens per frame
    (mdl_vog.py:538-744). `mul_tx.cross_frm=True` is broken in the reference
    (KeyError at mdl_vog.py:660) and is rejected here."""
    kind = "vog"

    def set_args_mdl(self):
        VidGrnd.set_args_mdl(self)
        mt = self.cfg.mdl.mul_tx
        assert mt.one_frm or mt.cross_frm
        if mt.cross_frm or not mt.one_frm:
            raise NotImplementedError("mul_tx.cross_frm / one_frm=False: not runnable in the reference either")

    def build_conc_model(self):
        VidGrnd.build_conc_model(self)
        self._take(("mult_txf.", "pe_mul_sub_enc."))


class ImgGrnd_SEP(ConcSEP, ImgGrnd):
    pass


class ImgGrnd_TEMP(ConcTEMP, ImgGrnd):
    pass


class ImgGrnd_SPAT(ConcSPAT, ImgGrnd):
    pass


class VidGrnd_SEP(ConcSEP, VidGrnd):
    pass


class VidGrnd_TEMP(ConcTEMP, VidGrnd):
    pass


class VidGrnd_SPAT(ConcSPAT, VidGrnd):
    pass


class VOG_SEP(ConcSEP, VOGNet):
    pass


class VOG_TEMP(ConcTEMP, VOGNet):
    pass


class VOG_SPAT(ConcSPAT, VOGNet):
    pass
