"""Host-side (NumPy, global ``np.random`` order preserved) target and proposal creators."""
from . import anchor_target_creator as _atc
from . import proposal_creator as _pc
from . import proposal_target_creator as _ptc

AnchorTargetCreator = _atc.AnchorTargetCreator
ProposalCreator = _pc.ProposalCreator
ProposalTargetCreator = _ptc.ProposalTargetCreator

__all__ = ['AnchorTargetCreator', 'ProposalCreator', 'ProposalTargetCreator']
