"""Ansatz specifications: the shape contract of the wave functions on the hot path.

The reference describes an ansatz as a hydra YAML tree instantiated into haiku modules
(reference src/deepqmc/conf/ansatz/{default,ferminet,psiformer}.yaml).  Here the same
information is a plain dataclass; every field cites the YAML key it restates.  The four
BASELINE.json configs use `paulinet()` (= conf/ansatz/default.yaml, the reference's
"PauliNet"), `ferminet()`, `psiformer()`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple, Union

Hidden = Union[Tuple[str, int], Tuple[int, ...]]


@dataclass(frozen=True)
class MLPSpec:
    """hkext.MLP arguments (reference src/deepqmc/hkext.py:22-113)."""

    hidden_layers: Hidden = ('log', 1)
    bias: Union[bool, str] = True          # True | False | 'not_last'
    last_linear: bool = False
    activation: Optional[str] = 'tanh'     # 'tanh' | 'silu' | None
    init: str = 'default'                  # 'default' | 'ferminet'

    def dims(self, in_dim: int, out_dim: int) -> list:
        """Layer widths.  ('log', n): round(in^(1-q) * out^q), q = k/n (hkext.py:85-91)."""
        hl = self.hidden_layers
        if len(hl) == 2 and hl[0] == 'log':
            n = int(hl[1])
            return [round(in_dim ** (1 - k / n) * out_dim ** (k / n)) for k in range(1, n + 1)]
        return [*[int(h) for h in hl], out_dim]

    def layer_bias(self, idx: int, n_layers: int) -> bool:
        """hkext.py:95-97."""
        return self.bias is True or (self.bias == 'not_last' and idx < n_layers - 1)

    def layer_act(self, idx: int, n_layers: int) -> Optional[str]:
        """hkext.py:110-112: activation on all but a `last_linear` last layer."""
        if idx < n_layers - 1 or not self.last_linear:
            return self.activation
        return None


@dataclass(frozen=True)
class AnsatzSpec:
    name: str
    # ---- NeuralNetworkWaveFunction (wf/nn_wave_function.py:77-113) ----
    n_determinants: int = 16               # n_determinants
    full_determinant: bool = True          # full_determinant
    conf_coeff: str = 'linear'             # 'linear' = hk.Linear(1,no bias,w=1) | 'sum' = SumPool
    # cusp_electrons (wf/cusp.py:49-78): None | 'deepqmc' | 'psiformer'
    cusp: Optional[str] = 'deepqmc'
    cusp_same_scale: float = 0.25
    cusp_anti_scale: float = 0.5
    cusp_alpha: float = 10.0
    cusp_trainable_alpha: bool = False
    # ---- OmniNet (wf/omni.py:91-178) ----
    embedding_dim: int = 128
    jastrow: Optional[MLPSpec] = None      # jastrow_factory.subnet_factory (sum_first: true)
    backflow: MLPSpec = field(default_factory=lambda: MLPSpec(('log', 1), False, True, None))
    # ---- ElectronGNN (gnn/electron_gnn.py:279-432) ----
    n_interactions: int = 3
    two_particle_dim: int = 32
    self_interaction: bool = False
    edge_types: Tuple[str, ...] = ('same', 'anti')   # keys of edge_features
    edge_log_rescale: bool = False
    # electron_embedding (gnn/electron_gnn.py:540-625): positional 'ne' features
    emb_log_rescale: bool = False
    emb_use_spin: bool = False
    emb_project: bool = False
    # ---- ElectronGNNLayer (gnn/electron_gnn.py:14-276) ----
    layer_kind: str = 'message_passing'    # 'message_passing' | 'attention'
    update_features: Tuple[str, ...] = ('residual', 'node_up', 'node_down', 'conv_same', 'conv_anti')
    g: MLPSpec = field(default_factory=lambda: MLPSpec(('log', 1), False, False, 'tanh'))
    u: Optional[MLPSpec] = field(default_factory=lambda: MLPSpec(('log', 2), True, False, 'tanh'))
    w: Optional[MLPSpec] = field(default_factory=lambda: MLPSpec(('log', 2), True, False, 'tanh'))
    h: Optional[MLPSpec] = field(default_factory=lambda: MLPSpec(('log', 2), True, False, 'tanh'))
    electron_residual_normalize: Optional[bool] = True   # None = no residual
    two_particle_residual_normalize: Optional[bool] = True
    # attention layers (gnn/update_features.py:214-286)
    num_heads: int = 4
    attn_mlp: Optional[MLPSpec] = None
    init: str = 'default'
    # ---- nuclear tokens (TransPsiformer: gnn/electron_gnn.py:435-537, update_features.py:385-451,
    # wf/omni.py:181-211, wf/env.py:110-226) ----
    nuclei_tokens: bool = False            # CombinedNodeAttention over [nuclei; electrons], elec_to_nuc = False
    nuc_types: Tuple[int, ...] = ()        # index of each nucleus' charge among the sorted unique charges
    envelope: str = 'exponential'          # 'exponential' (env.py:57-108) | 'simplified' (env.py:110-226)
    n_envelope_per_nucleus: int = 1

    @property
    def deep_features(self) -> bool:
        return self.u is not None


def paulinet() -> AnsatzSpec:
    """conf/ansatz/default.yaml (the reference's PauliNet-like default ansatz)."""
    return AnsatzSpec(
        name='paulinet',
        jastrow=MLPSpec(('log', 1), False, True, None, 'default'),      # default.yaml:40-50
        backflow=MLPSpec(('log', 1), False, True, None, 'default'),     # default.yaml:51-61
    )


def ferminet() -> AnsatzSpec:
    """conf/ansatz/ferminet.yaml."""
    return AnsatzSpec(
        name='ferminet',
        conf_coeff='sum', cusp=None,                                      # ferminet.yaml:18-23
        embedding_dim=256, jastrow=None,
        backflow=MLPSpec(('log', 1), False, True, None, 'ferminet'),
        n_interactions=4, two_particle_dim=32, self_interaction=True,
        edge_types=('up', 'down'),
        update_features=('residual', 'node_up', 'node_down', 'edge_up', 'edge_down'),
        g=MLPSpec(('log', 1), True, False, 'tanh', 'ferminet'),         # subnet_factory
        u=MLPSpec(('log', 1), True, False, 'tanh', 'ferminet'),
        w=None, h=None, init='ferminet',
    )


def psiformer() -> AnsatzSpec:
    """conf/ansatz/psiformer.yaml."""
    return AnsatzSpec(
        name='psiformer',
        conf_coeff='sum',
        cusp='psiformer', cusp_alpha=1.0, cusp_trainable_alpha=True,     # psiformer.yaml:18-26
        embedding_dim=256, jastrow=None,
        backflow=MLPSpec(('log', 1), False, True, None, 'ferminet'),
        n_interactions=4, two_particle_dim=32, self_interaction=True,
        edge_types=(), emb_log_rescale=True, emb_use_spin=True, emb_project=True,
        layer_kind='attention', update_features=('attention',),
        g=None, u=None, w=None, h=None,
        electron_residual_normalize=None, two_particle_residual_normalize=None,
        num_heads=4,
        attn_mlp=MLPSpec(('log', 2), True, False, 'tanh', 'ferminet'),
        init='ferminet',
    )


def transpsiformer(charges=None) -> AnsatzSpec:
    """conf/ansatz/transpsiformer.yaml: Psiformer whose attention also runs over nuclear tokens (which do not
    see the electrons: `elec_to_nuc: false`) and whose envelope exponents are read out from the final nuclear
    embeddings.  `charges` fixes the atom types of the nuclei embedding (one-hot over the sorted unique charges,
    electron_gnn.py:497-503); NeuralNetworkWaveFunction fills it in from the Hamiltonian."""
    import dataclasses
    types = ()
    if charges is not None:
        uniq = sorted(set(float(c) for c in charges))
        types = tuple(uniq.index(float(c)) for c in charges)
    return dataclasses.replace(psiformer(), name='transpsiformer', nuclei_tokens=True, nuc_types=types,
                               envelope='simplified', n_envelope_per_nucleus=3)


ANSATZES = {'paulinet': paulinet, 'default': paulinet, 'ferminet': ferminet, 'psiformer': psiformer,
            'transpsiformer': transpsiformer}
